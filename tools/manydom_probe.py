#!/usr/bin/env python3
"""pcg on the 160-subdomain problem (n_Γ = 9417): time per iteration of the folded loop beyond FUSED_MAX_N and of the
generic loop (MI355_NO_BIG_FOLD=1), for the tile shapes MI355_GEMV_RPW / MI355_GEMV_WAVES given in the environment."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package()
fem, api = pkg.fem, pkg.api
import torch
N, px, py = 400, 16, 10
mesh = fem.get_mesh(N)
g = fem.draw(fem.synthetic_kl(mesh.points), np.random.default_rng(481456))[1]
P = fem.build_schur_problem(N, px, py, np.exp(g), lambda x, y: -1.0 + 0 * x, lambda x, y: 0.734 + 0 * x, mesh=mesh)
ctx = api.Context(0)
n = P.sub.n_Γ
bd = torch.from_numpy(P.b_schur).cuda()
z = lambda: torch.zeros(n, dtype=torch.float64, device="cuda")
for label, env in (("folded loop (32-row tiles)", {}), ("folded, 8 waves x 2 rows", {"MI355_GEMV_WAVES": "8"}), ("generic loop", {"MI355_NO_BIG_FOLD": "1"})):
    os.environ.update(env)
    S = api.LocalSchurs(ctx, P.Sd, P.sub.gather_idx, P.sub.node_Γ_cnt)
    M = api.NeumannNeumannSchurPreconditioner(ctx, P.ΠSd, P.sub.gather_idx, P.sub.node_Γ_cnt)
    def t(maxit):
        api.pcg(S, bd, z(), M, maxit=maxit); api.pcg(S, bd, z(), M, maxit=maxit)
        t0 = time.perf_counter()
        for _ in range(5):
            it = api.pcg(S, bd, z(), M, maxit=maxit)[1]
        return (time.perf_counter() - t0) / 5, it
    tf, itf = t(0)
    ts, its = t(50)
    print(f"{label:46s}: it={itf} {tf * 1e3:7.2f} ms/solve, {(tf - ts) / (itf - its) * 1e6:6.2f} us/iteration", flush=True)
    for k in env: del os.environ[k]
