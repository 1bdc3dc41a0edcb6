#!/usr/bin/env python3
"""Sweep the batched-GEMV kernel variants (waves per workgroup, rows per wave) on the 1M-DoF blocks and
report us/launch and algorithmic GB/s for the S-apply and the NN-apply, plus PCG time per solve."""
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

N = int(os.environ.get("SWEEP_N", "1000"))
mesh = fem.get_mesh(N)
_, g = fem.draw(fem.synthetic_kl(mesh.points), np.random.default_rng(481456))
P = fem.build_schur_problem(N, 4, 2, np.exp(g), lambda x, y: -1.0 + 0 * x, lambda x, y: 0.734 + 0 * x)
ctx = api.Context(0)
b = torch.from_numpy(P.b_schur).cuda()
n = P.sub.n_Γ
for waves in (4, 8, 16):
    for rpw in (1, 2, 4):
        os.environ["MI355_GEMV_WAVES"], os.environ["MI355_GEMV_RPW"] = str(waves), str(rpw)
        S = api.LocalSchurs(ctx, P.Sd, P.sub.gather_idx, P.sub.node_Γ_cnt)
        M = api.NeumannNeumannSchurPreconditioner(ctx, P.ΠSd, P.sub.gather_idx, P.sub.node_Γ_cnt)
        out = []
        for op in (S, M):
            _, nb = op.bytes()
            us = op.time_dominant(b, 300)
            out.append((us, nb / us / 1e3))
        for _ in range(5):
            x, it, res = api.pcg(S, b, torch.zeros_like(b), M)
        xs = [torch.zeros_like(b) for _ in range(100)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(100):
            api.pcg(S, b, xs[k], M)
        dt = (time.perf_counter() - t0) / 100
        print(f"waves={waves} rpw={rpw}: S {out[0][0]:6.2f} us {out[0][1]:7.1f} GB/s | NN {out[1][0]:6.2f} us {out[1][1]:7.1f} GB/s"
              f" | pcg it={it} {dt * 1e6:7.1f} us/solve {(it - 1) / dt:8.0f} it/s", flush=True)
        del S, M
