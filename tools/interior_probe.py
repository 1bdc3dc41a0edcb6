#!/usr/bin/env python3
"""Matrix-free S-apply at config 3 with the interior CG on the device: the 3-launch loop (default) and the 2-launch loop
(MI355_ICG_FUSED=1), each plain and with the diagonal `Pl`. Prints time per apply, iterations and the
time per interior iteration."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
fem, api = pkg.fem, pkg.api

N = int(os.environ.get("MEASURE_N", "1000"))
mesh = fem.get_mesh(N)
_, g = fem.draw(fem.synthetic_kl(mesh.points), np.random.default_rng(481456))
P = fem.build_schur_problem(N, 4, 2, np.exp(g), lambda x, y: -1.0 + 0 * x, lambda x, y: 0.734 + 0 * x)
ctx = api.Context(0)
n, b = P.sub.n_Γ, P.b_schur
S = api.LocalSchurs(ctx, P.Sd, P.sub.gather_idx, P.sub.node_Γ_cnt)
ya = S * b
args = (ctx, P.A_IIdd, P.A_IΓdd, P.A_ΓΓdd, P.sub.gather_idx, P.sub.node_Γ_cnt, None)


def probe(label, op):
    op * b
    i0 = op.interior_iterations()
    t0 = time.perf_counter()
    y = op * b
    dt = time.perf_counter() - t0
    its = op.interior_iterations() - i0
    print(f"{label:38s}: {dt * 1e3:8.1f} ms/apply, {its} interior iterations (replay granularity), "
          f"{dt / max(1, its) * 1e6:6.1f} us/iteration, |Δ| vs assembled = {np.abs(y - ya).max() / np.abs(ya).max():.2e}", flush=True)


probe("3-launch interior CG (default)", api.MatrixFreeLocalSchurs(*args, reltol=1e-9))
Sj = api.MatrixFreeLocalSchurs(*args, reltol=1e-9)
Sj.interior_precond("diagonal")
probe("3-launch interior CG, Pl = diagonal", Sj)
os.environ["MI355_ICG_FUSED"] = "1"
probe("2-launch interior CG (MI355_ICG_FUSED=1)", api.MatrixFreeLocalSchurs(*args, reltol=1e-9))
Sj2 = api.MatrixFreeLocalSchurs(*args, reltol=1e-9)
del os.environ["MI355_ICG_FUSED"]
Sj2.interior_precond("diagonal")
probe("2-launch interior CG, Pl = diagonal", Sj2)
