#!/usr/bin/env python3
"""The persistent on-chip PCG (csrc/resident.hpp) against the folded graph loop and the oracle: iteration counts,
histories, solutions, time per solve. `--N 1000` is config 3."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # (lives under tests/: it uses the oracle as its checker)
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, default=1000)
ap.add_argument("--px", type=int, default=4)
ap.add_argument("--py", type=int, default=2)
ap.add_argument("--reps", type=int, default=100)
ap.add_argument("--small", action="store_true", help="only the small cases")
args = ap.parse_args()
pkg = graft.load_package()
fem, api = pkg.fem, pkg.api
from oracle import oracle as orc  # noqa: E402
import torch  # noqa: E402

ctx = api.Context(0)


def run(P, label, reps):
    n = P.sub.n_Γ
    S = api.LocalSchurs(ctx, P.Sd, P.sub.gather_idx, P.sub.node_Γ_cnt)
    M = api.NeumannNeumannSchurPreconditioner(ctx, P.ΠSd, P.sub.gather_idx, P.sub.node_Γ_cnt)
    So = orc.apply_local_schurs_operator(P.Sd, P.sub.gather_idx, n)
    Mo = orc.neumann_neumann_operator(P.ΠSd, P.sub.gather_idx, P.sub.node_Γ_cnt)
    xo, ito, reso = orc.pcg(So, P.b_schur, np.zeros(n), Mo)
    out = {}
    for mode in ("resident", "folded"):
        os.environ["MI355_RESIDENT"] = "1" if mode == "resident" else "0"
        x, it, res = api.pcg(S, P.b_schur, np.zeros(n), M)
        dev = np.max(np.abs(res[:min(it, ito)] - reso[:min(it, ito)]) / reso[:min(it, ito)])
        b = torch.from_numpy(P.b_schur).cuda()
        xs = [torch.zeros(n, dtype=torch.float64, device="cuda") for _ in range(reps)]
        for k in range(3):
            api.pcg(S, b, torch.zeros_like(b), M)
        ctx.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(reps):
            api.pcg(S, b, xs[k], M)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out[mode] = (x, it, res)
        print(f"{label:28s} {mode:9s}: it={it} (oracle {ito})  max rel dev of res_norm {dev:.2e}  |x-xo|/|xo| "
              f"{np.linalg.norm(x - xo) / np.linalg.norm(xo):.2e}  {dt * 1e6:8.1f} us/solve  {(it - 1) / dt:9.0f} it/s", flush=True)
    # a non-zero initial guess and a capped solve
    os.environ["MI355_RESIDENT"] = "1"
    x0 = np.random.default_rng(1).standard_normal(n)
    g = api.pcg(S, P.b_schur, x0, M); w = orc.pcg(So, P.b_schur, x0, Mo)
    print(f"{'':28s} x0 != 0  : it={g[1]} (oracle {w[1]})  |x-xo|/|xo| {np.linalg.norm(g[0] - w[0]) / np.linalg.norm(w[0]):.2e}", flush=True)
    g = api.pcg(S, P.b_schur, np.zeros(n), M, maxit=4); w = orc.pcg(So, P.b_schur, np.zeros(n), Mo, maxit=4)
    print(f"{'':28s} maxit=4  : it={g[1]} (oracle {w[1]})  |x-xo|/|xo| {np.linalg.norm(g[0] - w[0]) / np.linalg.norm(w[0]):.2e}", flush=True)
    os.environ.pop("MI355_RESIDENT")


one = lambda x, y: 1.0 + 0 * x      # noqa: E731
f = lambda x, y: -1.0 + 0 * x       # noqa: E731
ue = lambda x, y: 0.734 + 0 * x     # noqa: E731
run(fem.build_schur_problem(40, 2, 2, one, f, ue), "micro N=40 2x2", 20)
mesh = fem.get_mesh(50)
_, g7 = fem.draw(fem.synthetic_kl(mesh.points), np.random.default_rng(7))
run(fem.build_schur_problem(50, 3, 2, np.exp(g7), f, ue), "ragged N=50 3x2", 20)
run(fem.build_schur_problem(100, 2, 2, one, f, ue), "toy N=100 2x2", 20)
if not args.small:
    mesh = fem.get_mesh(args.N)
    _, g = fem.draw(fem.synthetic_kl(mesh.points), np.random.default_rng(481456))
    run(fem.build_schur_problem(args.N, args.px, args.py, np.exp(g), f, ue), f"N={args.N} {args.px}x{args.py}", args.reps)
