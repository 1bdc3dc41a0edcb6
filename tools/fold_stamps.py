#!/usr/bin/env python3
"""Wall-clock stamps (s_memrealtime, 100 MHz) of ONE workgroup of the folded PCG launches over one config-3 solve:
   MI355_FOLD_DEBUG=1 MI355_FOLD_DEBUG_WG=<tile> python tools/fold_stamps.py
Columns per launch (us since the first stamp): entry, prologue loads issued, scalars reduced, operand staged, stream consumed,
results scattered, exit. Launch 2k is the ΠS phase of iteration k, 2k+1 the S phase."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MI355_FOLD_DEBUG", "1")
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
fem, api = pkg.fem, pkg.api
import torch  # noqa: E402

N = int(os.environ.get("MEASURE_N", "1000"))
px, py = int(os.environ.get("MEASURE_PX", "4")), int(os.environ.get("MEASURE_PY", "2"))
mesh = fem.get_mesh(N)
_, g = fem.draw(fem.synthetic_kl(mesh.points), np.random.default_rng(481456))
P = fem.build_schur_problem(N, px, py, np.exp(g), lambda x, y: -1.0 + 0 * x, lambda x, y: 0.734 + 0 * x)
ctx = api.Context(0)
S = api.LocalSchurs(ctx, P.Sd, P.sub.gather_idx, P.sub.node_Γ_cnt)
M = api.NeumannNeumannSchurPreconditioner(ctx, P.ΠSd, P.sub.gather_idx, P.sub.node_Γ_cnt)
bd = torch.from_numpy(P.b_schur).cuda()
print("n_Γd", [len(a) for a in P.sub.gather_idx], file=sys.stderr)
for k in range(4):   # the stamps of every solve are printed by the library; the last ones are the steady state
    print(f"--- solve {k}", file=sys.stderr, flush=True)
    api.pcg(S, bd, torch.zeros_like(bd), M, maxit=int(os.environ.get("MEASURE_MAXIT", "0")))
