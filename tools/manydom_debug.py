#!/usr/bin/env python3
"""Debug aid: the eigCG family on the 160-subdomain problem (generic loop), bounded by maxit."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package()
fem, api = pkg.fem, pkg.api
from oracle import oracle as orc
orc.build()
N, px, py = 400, 16, 10
mesh = fem.get_mesh(N)
g = fem.draw(fem.synthetic_kl(mesh.points), np.random.default_rng(481456))[1]
P = fem.build_schur_problem(N, px, py, np.exp(g), lambda x, y: -1.0 + 0 * x, lambda x, y: 0.734 + 0 * x, mesh=mesh)
ctx = api.Context(0)
n, b = P.sub.n_Γ, P.b_schur
S = api.LocalSchurs(ctx, P.Sd, P.sub.gather_idx, P.sub.node_Γ_cnt)
M = api.NeumannNeumannSchurPreconditioner(ctx, P.ΠSd, P.sub.gather_idx, P.sub.node_Γ_cnt)
So = orc.apply_local_schurs_operator(P.Sd, P.sub.gather_idx, n)
Mo = orc.neumann_neumann_operator(P.ΠSd, P.sub.gather_idx, P.sub.node_Γ_cnt)
for maxit in (30, 100, 0):
    t0 = time.time()
    try:
        x, it, res, W = api.eigpcg(S, b, np.zeros(n), M, 20, 60, maxit=maxit)
        msg = f"it={it} res[-1]={res[-1]:.3e}"
    except Exception as e:
        msg = f"{type(e).__name__}: {e}"
    t1 = time.time()
    try:
        xo, ito, reso, Wo = orc.eigpcg(So, b, np.zeros(n), Mo, 20, 60, maxit=maxit)
        msgo = f"it={ito} res[-1]={reso[-1]:.3e}"
    except Exception as e:
        msgo = f"{type(e).__name__}: {e}"
    print(f"maxit={maxit}: gpu {msg} ({t1 - t0:.2f} s) | oracle {msgo}", flush=True)
    if maxit and 'res' in dir():
        k = min(len(res), len(reso))
        print("   first divergence:", next((i for i in range(k) if abs(res[i] - reso[i]) > 1e-6 * reso[i]), None), flush=True)

xo, ito, reso, Wo = orc.eigpcg(So, b, np.zeros(n), Mo, 20, 60)
for maxit in (10, 40, 0):
    t0 = time.time()
    try:
        x, it, res, W2 = api.eigdefpcg(S, b, np.zeros(n), M, Wo, 60, maxit=maxit)
        msg = f"it={it} res[-1]={res[-1]:.3e}"
    except Exception as e:
        msg = f"{type(e).__name__}: {e}"
    t1 = time.time()
    try:
        x2, it2, res2, _ = orc.eigdefpcg(So, b, np.zeros(n), Mo, Wo, 60, maxit=maxit)
        msgo = f"it={it2} res[-1]={res2[-1]:.3e}"
    except Exception as e:
        msgo = f"{type(e).__name__}: {e}"
    print(f"eigdefpcg maxit={maxit}: gpu {msg} ({t1 - t0:.2f} s) | oracle {msgo}", flush=True)
