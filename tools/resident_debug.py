#!/usr/bin/env python3
"""One pcg solve at config 3 with MI355_RES_DEBUG=1: prints the wall-clock stamps of one workgroup of the persistent kernel."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
fem, api = pkg.fem, pkg.api
N = int(os.environ.get("PROBE_N", "1000"))
mesh = fem.get_mesh(N)
_, g = fem.draw(fem.synthetic_kl(mesh.points), np.random.default_rng(481456))
P = fem.build_schur_problem(N, 4, 2, np.exp(g), lambda x, y: -1.0 + 0 * x, lambda x, y: 0.734 + 0 * x)
ctx = api.Context(0)
n = P.sub.n_Γ
S = api.LocalSchurs(ctx, P.Sd, P.sub.gather_idx, P.sub.node_Γ_cnt)
M = api.NeumannNeumannSchurPreconditioner(ctx, P.ΠSd, P.sub.gather_idx, P.sub.node_Γ_cnt)
os.environ["MI355_RESIDENT"] = "1"
for k in range(3):
    api.pcg(S, P.b_schur, np.zeros(n), M)
for wg in (0, 100, 255):
    os.environ["MI355_RES_DEBUG"] = "1"
    os.environ["MI355_RES_DEBUG_WG"] = str(wg)
    x, it, res = api.pcg(S, P.b_schur, np.zeros(n), M)
    print("it", it, flush=True)
