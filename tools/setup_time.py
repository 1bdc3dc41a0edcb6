#!/usr/bin/env python3
"""Time of one device set-up (mi_schur_setup_run) at config 3, nothing else: for A/B builds of csrc/setup_gj.hpp
(MI355SCHUR_LIB selects the library)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_package(); fem, api = pkg.fem, pkg.api
import torch
N = int(os.environ.get("PROBE_N", "1000"))
mesh = fem.get_mesh(N)
_, g = fem.draw(fem.synthetic_kl(mesh.points), np.random.default_rng(481456))
P = fem.build_schur_problem(N, 4, 2, np.exp(g), lambda x, y: -1.0 + 0 * x, lambda x, y: 0.734 + 0 * x, assemble=False, precond=False)
ctx = api.Context(0)
setup = api.SchurSetup(ctx, P.A_IIdd, P.A_IΓdd, P.A_ΓΓdd)
vals = [torch.from_numpy(v).cuda() for v in setup._vals]
bI = torch.from_numpy(np.concatenate(P.b_Id)).cuda()
setup.run(*vals, bI); ctx.synchronize()
ts = []
for _ in range(4):
    t0 = time.perf_counter(); Sd, w = setup.run(*vals, bI); ctx.synchronize(); ts.append(time.perf_counter() - t0)
print(f"{os.environ.get('MI355SCHUR_LIB', 'default library')}: {min(ts) * 1e3:.1f} ms per realization; finite: {bool(torch.isfinite(Sd).all())}", flush=True)
