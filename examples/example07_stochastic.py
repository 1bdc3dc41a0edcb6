#!/usr/bin/env python3
"""Example07_PcgSchurStochasticEllipticPde.jl (lines 56-285) on the MI355X drop-in — BASELINE config 5.

For every realization ξ_t of the lognormal coefficient: rebuild the local blocks, b_schur, the assembled
S_d and NN_t on the host (as the reference does), then on the GPU
    pcg(S, b_schur, 0, ΠSnn_0)        (Example07:273)   preconditioner of the ξ = 0 operator, built once
    pcg(S, b_schur, 0, ΠSnn_t)        (Example07:277)   preconditioner of this realization
    defpcg(S, b_schur, 0, W_0, ΠSnn_0)                  deflation with the nev least-dominant eigenvectors of
                                                        S_0 (the `defpcg` variant that Example07 keeps in its
                                                        trailing comment block, with W fixed instead of recycled)
    eigpcg / eigdefpcg(S, b_schur, 0, ΠSnn_0, W, spdim)   (--recycle) BASELINE config 5's "deflated Schur-PCG with
                                                        recycled W": the first system runs eigpcg, every later one
                                                        eigdefpcg with the W the previous solve returned
                                                        (Example09_..._Functions.jl:345, 364; nvec = 1.25 ndom,
                                                        spdim = 3 ndom), the chain being per rank
and record the iteration counts. Realizations are independent: under torch.distributed.run each rank takes
realizations rank, rank+world, ... on its own GPU (replicas only, no collective in the solve).

With --device-assembly the element loop of `prepare_local_schurs` (Example07:162-171, redone per realization by the
reference) runs on the GPU: the index half is prepared once (`fem.make_assembly_plan`), each realization is one
`mi_assembly_run` (0.2 ms at 1 M DoF vs ~1-3 s for the host loop); the block values come back for the dense
elimination that builds S_d.

    python examples/example07_stochastic.py [--N 200 --px 4 --py 2 --nreals 20 --device-assembly]
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
    ap.add_argument("--nreals", type=int, default=20)   # Example07:29 nreals = 1000
    ap.add_argument("--seed", type=int, default=481456)
    ap.add_argument("--out", default="")
    ap.add_argument("--device-assembly", action="store_true")
    ap.add_argument("--device-setup", action="store_true",
                    help="with --device-assembly: S_d, the condensed rhs and NN_t on the device too (mi_schur_setup_run, mi_nn_pinv, "
                         "mi_dense_set_blocks): nothing of a realization's set-up runs on the host")
    ap.add_argument("--recycle", action="store_true")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    pkg = graft.load_package()
    fem, api = pkg.fem, pkg.api
    f = lambda x, y: -1.0 + 0 * x
    uexact = lambda x, y: 0.734 + 0 * x
    mesh = fem.get_mesh(args.N)
    kl = fem.synthetic_kl(mesh.points)
    rng = np.random.default_rng(args.seed)
    gs = [fem.draw(kl, rng)[1] for _ in range(args.nreals)]          # Example07:140-144: all draws up front

    ctx = api.Context(int(os.environ.get("LOCAL_RANK", "0")))
    P0 = fem.build_schur_problem(args.N, args.px, args.py, np.exp(0 * gs[0]), f, uexact)   # ξ = 0 operator (:88-137)
    sub = P0.sub
    n_Γ, ndom = sub.n_Γ, sub.ndom
    ΠSnn_0 = api.NeumannNeumannSchurPreconditioner(ctx, P0.ΠSd, sub.gather_idx, sub.node_Γ_cnt)      # :152-154
    S_0 = api.LocalSchurs(ctx, P0.Sd, sub.gather_idx, sub.node_Γ_cnt)
    S0d = np.column_stack([S_0 * e for e in np.eye(n_Γ)])
    W_0 = np.asfortranarray(np.linalg.eigh((S0d + S0d.T) / 2)[1][:, :ndom + 10])

    plan = dev_plan = None
    if args.device_assembly:
        plan = fem.make_assembly_plan(mesh.cells, mesh.points, P0.epart, sub, f, uexact)
        dev_plan = api.AssemblyPlan(ctx, plan)
    setup = S_dev = NN_dev = None
    if args.device_setup:
        if not args.device_assembly:
            raise SystemExit("--device-setup needs --device-assembly")
        import torch
        setup = api.SchurSetup(ctx, P0.A_IIdd, P0.A_IΓdd, P0.A_ΓΓdd)
        S_dev = api.LocalSchurs(ctx, P0.Sd, sub.gather_idx, sub.node_Γ_cnt)
        NN_dev = api.NeumannNeumannSchurPreconditioner(ctx, P0.ΠSd, sub.gather_idx, sub.node_Γ_cnt)
    iters_0, iters_t, iters_def, iters_rec = [], [], [], []
    W_rec, nvec, spdim = None, int(1.25 * ndom), 3 * ndom
    for ireal in range(rank, args.nreals, world):
        if setup is not None:
            # the whole realization on the device: element loop -> block values -> S_d, w_d -> operators refilled
            vals = dev_plan.run(torch.from_numpy(np.exp(gs[ireal])).cuda())
            ii, ig, gg, bI, bΓ = dev_plan.block_values(vals)
            Sd, w = setup.run(ii, ig, gg, bI)                                                          # :180-187
            S_dev.set_blocks(Sd)
            NN_dev.set_blocks(api.nn_pinv(ctx, sub.n_Γd, Sd))                                          # :190-199
            ctx.synchronize()
            b_schur, wh, off = bΓ.cpu().numpy().copy(), w.cpu().numpy(), 0
            for d in range(ndom):                                                                      # get_schur_rhs, EPDD.jl:853-861
                b_schur[sub.gather_idx[d]] -= wh[off:off + sub.n_Γd[d]]
                off += sub.n_Γd[d]
            S, ΠSnn_t = S_dev, NN_dev
        else:
            blocks = plan.blocks(dev_plan.run(np.exp(gs[ireal]))) if plan else None                   # :162-171 on the GPU
            P = fem.build_schur_problem(args.N, args.px, args.py, np.exp(gs[ireal]), f, uexact, mesh=mesh,
                                        partition=(P0.epart, None), blocks=blocks, sub=sub)      # :162-199
            S = api.LocalSchurs(ctx, P.Sd, sub.gather_idx, sub.node_Γ_cnt)
            ΠSnn_t = api.NeumannNeumannSchurPreconditioner(ctx, P.ΠSd, sub.gather_idx, sub.node_Γ_cnt)
            b_schur = P.b_schur
        x0 = np.zeros(n_Γ)
        iters_0.append(api.pcg(S, b_schur, x0, ΠSnn_0)[1])                                            # :273
        iters_t.append(api.pcg(S, b_schur, x0, ΠSnn_t)[1])                                            # :277
        iters_def.append(api.defpcg(S, b_schur, x0, W_0, ΠSnn_0)[1])
        rec = ""
        if args.recycle:
            try:
                if W_rec is None:
                    _, it, _, W_rec = api.eigpcg(S, b_schur, x0, ΠSnn_0, nvec, spdim)
                else:
                    _, it, _, W_rec = api.eigdefpcg(S, b_schur, x0, ΠSnn_0, W_rec, spdim)
                iters_rec.append(it)
                rec = f"  eig(def)pcg(W recycled, NN_0) it={it}"
            except (api.BoundsError, api.SingularException) as e:          # Example09:355-375: status = -1
                rec = f"  recycling stopped: {type(e).__name__}"
                W_rec = None
        print(f"[rank {rank}] realization {ireal}: pcg(NN_0) it={iters_0[-1]}  pcg(NN_t) it={iters_t[-1]}  "
              f"defpcg(W_0, NN_0) it={iters_def[-1]}{rec}", flush=True)
    if args.out:                                                                                       # :281-285 npz of iteration counts
        np.savez(args.out.format(rank=rank), iters_0=iters_0, iters_t=iters_t, iters_def=iters_def, iters_rec=iters_rec)
    print(f"[rank {rank}] mean its: NN_0 {np.mean(iters_0):.1f}  NN_t {np.mean(iters_t):.1f}  def {np.mean(iters_def):.1f}")


if __name__ == "__main__":
    main()
