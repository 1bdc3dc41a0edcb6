#!/usr/bin/env python3
"""The recycling chain of Example09_DefPcgMcmcStochasticEllipticPde_Functions.jl:295-377 on the MI355X drop-in.

A sequence of CORRELATED realizations of the lognormal coefficient (the reference walks an MCMC chain; here an AR(1)
walk ξ_{s+1} = ρ ξ_s + sqrt(1-ρ²) η with the same synthetic KL modes — the sampler itself is out of scope) is solved
on the NN-preconditioned Schur system with the preconditioner of the ξ = 0 operator (Example09: `neumann-neumann_0`):

    s = 1:  eigpcg(S, b_schur, 0, ΠSnn_0, nvec, spdim)                -> W            (:345)
    s > 1:  eigdefpcg(S_s, b_schur_s, 0, ΠSnn_0, W, spdim)            -> W            (:364)
    and, for comparison, pcg(S_s, b_schur_s, 0, ΠSnn_0)                               (:281)

with nvec = floor(1.25 ndom), spdim = 3 ndom (Example09:39-40). BoundsError / SingularException from the solver are
caught and recorded as status -1, as the reference does (:286-289, 355-375).

    python examples/example09_recycling.py [--N 200 --px 4 --py 2 --nsmp 10 --rho 0.95]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=200)
    ap.add_argument("--px", type=int, default=4)
    ap.add_argument("--py", type=int, default=2)
    ap.add_argument("--nsmp", type=int, default=10)
    ap.add_argument("--rho", type=float, default=0.95)
    ap.add_argument("--seed", type=int, default=481456)     # Example09:52
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    pkg = graft.load_package()
    fem, api = pkg.fem, pkg.api
    f = lambda x, y: -1.0 + 0 * x
    uexact = lambda x, y: 0.734 + 0 * x
    mesh = fem.get_mesh(args.N)
    kl = fem.synthetic_kl(mesh.points)
    rng = np.random.default_rng(args.seed)
    ctx = api.Context(int(os.environ.get("LOCAL_RANK", "0")))

    P0 = fem.build_schur_problem(args.N, args.px, args.py, np.ones(mesh.points.shape[1]), f, uexact)
    sub = P0.sub
    n_Γ, ndom = sub.n_Γ, sub.ndom
    nvec, spdim = int(1.25 * ndom), 3 * ndom
    ΠSnn_0 = api.NeumannNeumannSchurPreconditioner(ctx, P0.ΠSd, sub.gather_idx, sub.node_Γ_cnt)

    ξ = rng.standard_normal(kl.Λ.size)
    W = None
    iters = {"pcg": [], "eigdefpcg": []}
    status = 0
    for s in range(1, args.nsmp + 1):
        g = (kl.Ψ * (np.sqrt(kl.Λ) * ξ)[None, :]).sum(axis=1)
        P = fem.build_schur_problem(args.N, args.px, args.py, np.exp(g), f, uexact, precond=False)
        S = api.LocalSchurs(ctx, P.Sd, sub.gather_idx, sub.node_Γ_cnt)
        it_pcg = api.pcg(S, P.b_schur, np.zeros(n_Γ), ΠSnn_0)[1]
        try:
            if W is None:
                _, it, _, W = api.eigpcg(S, P.b_schur, np.zeros(n_Γ), ΠSnn_0, nvec, spdim)
            else:
                _, it, _, W = api.eigdefpcg(S, P.b_schur, np.zeros(n_Γ), ΠSnn_0, W, spdim)
        except (api.BoundsError, api.SingularException) as e:
            print(f"sample {s}: {type(e).__name__}: status = -1", flush=True)
            status = -1
            break
        iters["pcg"].append(it_pcg)
        iters["eigdefpcg"].append(it)
        print(f"sample {s}: pcg(NN_0) it={it_pcg}   {'eigpcg' if s == 1 else 'eigdefpcg'}(NN_0) it={it}", flush=True)
        ξ = args.rho * ξ + np.sqrt(1 - args.rho ** 2) * rng.standard_normal(ξ.size)
    if args.out:
        np.savez(args.out, **{k: np.array(v) for k, v in iters.items()}, status=status)
    if iters["pcg"]:
        print(f"mean its: pcg {np.mean(iters['pcg']):.1f}   eigpcg/eigdefpcg {np.mean(iters['eigdefpcg']):.1f}")


if __name__ == "__main__":
    main()
