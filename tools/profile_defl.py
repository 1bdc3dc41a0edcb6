#!/usr/bin/env python3
"""Kernel-level look at the deflated / eig solvers at config 3.
  python tools/profile_defl.py save /tmp/p.npz            (builds the 1M-DoF problem once, outside the profiler)
  rocprofv3 --kernel-trace --stats ... -- python3 tools/profile_defl.py defpcg /tmp/p.npz"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package(); fem, api = pkg.fem, pkg.api
mode, path = sys.argv[1], sys.argv[2]
if mode == "save":
    N = int(os.environ.get("MEASURE_N", "1000"))
    mesh = fem.get_mesh(N)
    _, g = fem.draw(fem.synthetic_kl(mesh.points), np.random.default_rng(481456))
    P = fem.build_schur_problem(N, 4, 2, np.exp(g), lambda x, y: -1.0 + 0 * x, lambda x, y: 0.734 + 0 * x)
    d = {"b": P.b_schur, "cnt": P.sub.node_Γ_cnt, "ndom": P.sub.ndom}
    for k in range(P.sub.ndom):
        d[f"S{k}"], d[f"P{k}"], d[f"g{k}"] = P.Sd[k], P.ΠSd[k], P.sub.gather_idx[k]
    np.savez(path, **d)
    sys.exit(0)
import torch
z = np.load(path)
nd = int(z["ndom"])
gi = [z[f"g{k}"] for k in range(nd)]
ctx = api.Context(0)
S = api.LocalSchurs(ctx, [z[f"S{k}"] for k in range(nd)], gi, z["cnt"])
M = api.NeumannNeumannSchurPreconditioner(ctx, [z[f"P{k}"] for k in range(nd)], gi, z["cnt"])
bd = torch.from_numpy(z["b"]).cuda()
nvec, spdim = 10, 24
W = api.eigpcg(S, bd, torch.zeros_like(bd), M, nvec, spdim)[3]
ctx.synchronize()
for _ in range(int(os.environ.get("REPS", "20"))):
    if mode == "defpcg":
        r = api.defpcg(S, bd, torch.zeros_like(bd), W, M)
    elif mode == "eigpcg":
        r = api.eigpcg(S, bd, torch.zeros_like(bd), M, nvec, spdim)
    elif mode == "pcg":
        r = api.pcg(S, bd, torch.zeros_like(bd), M)
    else:
        r = api.eigdefpcg(S, bd, torch.zeros_like(bd), M, W, spdim)
print(mode, "it", r[1])
