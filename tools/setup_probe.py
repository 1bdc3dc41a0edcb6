#!/usr/bin/env python3
"""Device set-up of the assembled mode at config 3: time per realization for several stream counts, and which of the two
dense eliminations (device level recursion, host level recursion through torch) is closer to an independent reference
(S_d v computed with sparse direct interior solves)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
fem, api = pkg.fem, pkg.api
import torch  # noqa: E402

N = int(os.environ.get("PROBE_N", "1000"))
mesh = fem.get_mesh(N)
_, g = fem.draw(fem.synthetic_kl(mesh.points), np.random.default_rng(481456))
t0 = time.perf_counter()
P = fem.build_schur_problem(N, 4, 2, np.exp(g), lambda x, y: -1.0 + 0 * x, lambda x, y: 0.734 + 0 * x)
print(f"host build_schur_problem: {time.perf_counter() - t0:.1f} s", flush=True)
sub = P.sub
ctx = api.Context(0)
for lanes in (1,):
    os.environ["MI355_SETUP_STREAMS"] = str(lanes)
    t0 = time.perf_counter()
    setup = api.SchurSetup(ctx, P.A_IIdd, P.A_IΓdd, P.A_ΓΓdd)
    tp = time.perf_counter() - t0
    vals = [torch.from_numpy(v).cuda() for v in setup._vals]
    bI = torch.from_numpy(np.concatenate(P.b_Id)).cuda()
    t0 = time.perf_counter(); setup.run(*vals, bI); ctx.synchronize()
    print(f"  first (eager) run {time.perf_counter() - t0:.2f} s", flush=True)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        Sd, w = setup.run(*vals, bI)
        ctx.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f"lanes={lanes}: plan {tp:.2f} s, S_d + w_d {min(ts) * 1e3:.1f} ms per realization", flush=True)
t0 = time.perf_counter()
Pi = api.nn_pinv(ctx, sub.n_Γd, Sd); ctx.synchronize()
print(f"pinv: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
blocks = [b.cpu().numpy() for b in setup.blocks(Sd)]
rng = np.random.default_rng(0)
for d in range(sub.ndom):
    v = rng.standard_normal(sub.n_Γd[d])
    ref = P.A_ΓΓdd[d] @ v - P.A_IΓdd[d].T @ P.solvers[d](P.A_IΓdd[d] @ v)
    e_dev = np.abs(blocks[d] @ v - ref).max() / np.abs(ref).max()
    e_host = np.abs(P.Sd[d] @ v - ref).max() / np.abs(ref).max()
    print(f"subdomain {d}: |S v - ref| / |ref|  device {e_dev:.2e}   host(torch levels) {e_host:.2e}   max|S_dev - S_host| {np.abs(blocks[d] - P.Sd[d]).max():.2e}", flush=True)
print("levels per subdomain:", [len(fem._bfs_levels_from_interface(P.A_IIdd[d].tocsr(), np.flatnonzero(np.diff(P.A_IΓdd[d].tocsr().indptr) > 0))) for d in (0, 1)])
