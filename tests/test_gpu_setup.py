"""GPU suite: set-up of the assembled mode on the device — `mi_schur_setup_*` (assemble_local_schurs,
EPDD.jl:667-695, and the condensed right-hand side of get_schur_rhs, :853-861), `mi_nn_pinv`
(prepare_neumann_neumann_schur_precond, :1201-1220) and `mi_dense_set_blocks` — against the host mirror in fem.py
(SuperLU / numpy; `method="solves"` is the independent cross-check of the level elimination) and numpy's pinv."""
import numpy as np
import pytest

from conftest import f_m1, lognormal_coeff, u0734

pytestmark = pytest.mark.gpu


def _host_blocks(fem, P):
    """Independent host route: multi-RHS sparse direct solves (not the level recursion the device restates)."""
    return fem.assemble_local_schurs(P.A_IIdd, P.A_IΓdd, P.A_ΓΓdd, solvers=P.solvers, method="solves")


@pytest.mark.parametrize("N,px,py,seed", [(50, 3, 2, 7), (31, 3, 3, 2), (90, 4, 2, 5)])
def test_device_assemble_local_schurs_and_pinv(pkg, ctx, fem, N, px, py, seed):
    api = pkg.api
    mesh = fem.get_mesh(N)
    P = fem.build_schur_problem(N, px, py, lognormal_coeff(fem, mesh.points, seed), f_m1, u0734)
    sub = P.sub
    setup = api.SchurSetup(ctx, P.A_IIdd, P.A_IΓdd, P.A_ΓΓdd)
    Sd, w = setup.run(b_I=np.concatenate(P.b_Id))
    blocks = setup.blocks(Sd)
    want = _host_blocks(fem, P)
    off = 0
    for d in range(sub.ndom):
        scale = np.abs(want[d]).max()
        assert np.abs(blocks[d] - want[d]).max() <= 1e-10 * scale, (d, np.abs(blocks[d] - want[d]).max() / scale)
        assert np.array_equal(blocks[d], blocks[d].T)                         # `Symmetric(Array(...))`, EPDD.jl:692
        wd = P.A_IΓdd[d].T @ P.solvers[d](P.b_Id[d])                           # A_IΓdd' (A_IIdd \ b_Id)
        n = sub.n_Γd[d]
        assert np.abs(w[off:off + n] - wd).max() <= 1e-10 * max(np.abs(wd).max(), 1e-300)
        off += n
    # the Schur right-hand side assembled from the device's w_d equals the host flow's
    b_schur = P.b_Γ.copy()
    off = 0
    for d in range(sub.ndom):
        b_schur[sub.gather_idx[d]] -= w[off:off + sub.n_Γd[d]]
        off += sub.n_Γd[d]
    assert np.abs(b_schur - P.b_schur).max() <= 1e-10 * np.abs(P.b_schur).max()
    # 1-based CSC arrays (what the Julia shim passes) give the same bits
    Sd1, _ = api.SchurSetup(ctx, P.A_IIdd, P.A_IΓdd, P.A_ΓΓdd, index_base=1).run()
    assert np.array_equal(Sd1, Sd)
    # pinv(S_d, rtol = sqrt(eps)): against numpy's SVD-based pinv of the same blocks (a floating subdomain — the centre box
    # of a 3x3 partition — has the constant vector in its kernel: one singular value dropped on both sides)
    n_spectral = ctx.query("spectral_pinv")
    Pi = setup.blocks(api.nn_pinv(ctx, sub.n_Γd, Sd))
    # no block takes the eigen-decomposition (rocSOLVER): full-rank blocks are inverted, the floating subdomain of the
    # 3x3 partition goes through the rank-one shift S^+ = (S + α u u')^{-1} - u u'/α
    assert ctx.query("spectral_pinv") == n_spectral
    rtol = float(np.sqrt(np.finfo(float).eps))
    for d in range(sub.ndom):
        ref = np.linalg.pinv(blocks[d], rcond=rtol)
        assert np.abs(Pi[d] - ref).max() <= 1e-7 * np.abs(ref).max(), (d, np.abs(Pi[d] - ref).max() / np.abs(ref).max())
        s = np.linalg.svd(blocks[d], compute_uv=False)
        rank = int((s > rtol * s[0]).sum())
        assert np.linalg.matrix_rank(Pi[d], tol=1e-6 * np.abs(Pi[d]).max()) == rank
    if (px, py) == (3, 3):
        s = np.linalg.svd(blocks[4], compute_uv=False)
        assert s[-1] <= 1e-10 * s[0]                                           # the floating subdomain is singular


@pytest.mark.parametrize("case", ["micro", "ragged"])
def test_device_setup_against_the_reference_semantics(pkg, ctx, orc, fem, micro, ragged, case):
    """a12 / a13 pinned to the ORACLE, i.e. to the reference's own semantics: `assemble_local_schurs` (EPDD.jl:667-695) applies
    `apply_local_schur` to every unit vector with an UNPRECONDITIONED `IterativeSolvers.cg(...; reltol=1e-9)` interior solve
    and mirrors the upper triangle; `prepare_neumann_neumann_schur_precond` (:1201-1220) is `pinv(rtol=sqrt(eps))`. The device
    set-up solves the interiors exactly (deviation N2 of SURVEY.md §0): this test MEASURES that deviation — S_d within
    1e-8 max|S_d| of the reference-semantics blocks, ΠS_d likewise — and checks that it does not reach the solver: NN-PCG
    takes the same number of iterations with either set of blocks, on the device and in the oracle."""
    from test_gpu_parity import assert_history
    api = pkg.api
    P = micro if case == "micro" else ragged
    sub = P.sub
    n = sub.n_Γ
    Sd_ref, its = orc.assemble_local_schurs(P.A_IIdd, P.A_IΓdd, P.A_ΓΓdd, reltol=1e-9, return_iterations=True)
    Pi_ref = orc.prepare_neumann_neumann_schur_precond(Sd_ref)
    assert all(i > 0 for i in its)
    setup = api.SchurSetup(ctx, P.A_IIdd, P.A_IΓdd, P.A_ΓΓdd)
    Sd_dev, _ = setup.run()
    Sd = setup.blocks(Sd_dev)
    Pi = setup.blocks(api.nn_pinv(ctx, sub.n_Γd, Sd_dev))
    dev_S = max(np.abs(Sd[d] - Sd_ref[d]).max() / np.abs(Sd_ref[d]).max() for d in range(sub.ndom))
    dev_P = max(np.abs(Pi[d] - Pi_ref[d]).max() / np.abs(Pi_ref[d]).max() for d in range(sub.ndom))
    print(f"\n{case}: deviation N2 (exact interior solves vs cg to reltol 1e-9): S_d {dev_S:.2e}, pinv(S_d) {dev_P:.2e} (relative, max norm)")
    assert dev_S <= 1e-8, dev_S
    assert dev_P <= 1e-6, dev_P            # pinv amplifies by cond(S_d) restricted to the kept spectrum
    for d in range(sub.ndom):
        assert np.array_equal(Sd_ref[d], Sd_ref[d].T)
    # the solver does not see the difference: same `it`, histories to the usual bar
    So_ref = orc.apply_local_schurs_operator(Sd_ref, sub.gather_idx, n)
    Mo_ref = orc.neumann_neumann_operator(Pi_ref, sub.gather_idx, sub.node_Γ_cnt)
    want = orc.pcg(So_ref, P.b_schur, np.zeros(n), Mo_ref)
    S_dev = api.LocalSchurs(ctx, [np.asfortranarray(b) for b in Sd], sub.gather_idx, sub.node_Γ_cnt)
    M_dev = api.NeumannNeumannSchurPreconditioner(ctx, [np.asfortranarray(b) for b in Pi], sub.gather_idx, sub.node_Γ_cnt)
    got = api.pcg(S_dev, P.b_schur, np.zeros(n), M_dev)
    assert got[1] == want[1], (got[1], want[1])
    assert np.allclose(got[2], want[2], rtol=1e-6, atol=1e-10 * want[2][0])      # blocks differ by 1e-9: histories follow
    assert np.linalg.norm(got[0] - want[0]) <= 1e-6 * np.linalg.norm(want[0])
    S_r = api.LocalSchurs(ctx, Sd_ref, sub.gather_idx, sub.node_Γ_cnt)
    M_r = api.NeumannNeumannSchurPreconditioner(ctx, Pi_ref, sub.gather_idx, sub.node_Γ_cnt)
    assert_history(api.pcg(S_r, P.b_schur, np.zeros(n), M_r), want)             # same blocks on both sides: the tight bar


def test_realization_on_the_device_end_to_end(pkg, ctx, orc, fem):
    """Example07's per-realization set-up without the host (Example07:162-199): element loop (mi_assembly_run) -> S_d, w_d
    (mi_schur_setup_run) -> operator update (mi_dense_set_blocks) -> ΠS_t (mi_nn_pinv) -> pcg, everything on device
    tensors; against the host flow for the same coefficient."""
    import torch
    from test_gpu_parity import assert_history
    api = pkg.api
    N, px, py = 60, 3, 2
    mesh = fem.get_mesh(N)
    a0, a1 = lognormal_coeff(fem, mesh.points, 1), lognormal_coeff(fem, mesh.points, 2)
    P0 = fem.build_schur_problem(N, px, py, a0, f_m1, u0734)
    P1 = fem.build_schur_problem(N, px, py, a1, f_m1, u0734)
    sub = P0.sub
    n = sub.n_Γ
    plan = fem.make_assembly_plan(mesh.cells, mesh.points, P0.epart, sub, f_m1, u0734)
    dev_plan = api.AssemblyPlan(ctx, plan)
    setup = api.SchurSetup(ctx, P0.A_IIdd, P0.A_IΓdd, P0.A_ΓΓdd)
    S = api.LocalSchurs(ctx, P0.Sd, sub.gather_idx, sub.node_Γ_cnt)
    M = api.NeumannNeumannSchurPreconditioner(ctx, P0.ΠSd, sub.gather_idx, sub.node_Γ_cnt)
    vals = dev_plan.run(torch.from_numpy(a1).cuda())
    ii, ig, gg, bI, bΓ = dev_plan.block_values(vals)
    Sd, w = setup.run(ii, ig, gg, bI)
    S.set_blocks(Sd)
    M.set_blocks(api.nn_pinv(ctx, sub.n_Γd, Sd))
    ctx.synchronize()
    b_schur = bΓ.cpu().numpy().copy()
    wh, off = w.cpu().numpy(), 0
    for d in range(sub.ndom):
        b_schur[sub.gather_idx[d]] -= wh[off:off + sub.n_Γd[d]]
        off += sub.n_Γd[d]
    assert np.abs(b_schur - P1.b_schur).max() <= 1e-10 * np.abs(P1.b_schur).max()
    So = orc.apply_local_schurs_operator(P1.Sd, sub.gather_idx, n)
    Mo = orc.neumann_neumann_operator(P1.ΠSd, sub.gather_idx, sub.node_Γ_cnt)
    v = np.random.default_rng(3).standard_normal(n)
    ys, ym = S * v, M.ldiv(v)
    assert np.abs(ys - So * v).max() <= 1e-10 * np.abs(ys).max()
    assert np.abs(ym - Mo * v).max() <= 1e-7 * np.abs(ym).max()
    got = api.pcg(S, b_schur, np.zeros(n), M)
    want = orc.pcg(So, P1.b_schur, np.zeros(n), Mo)
    assert got[1] == want[1] and np.linalg.norm(got[0] - want[0]) <= 1e-6 * np.linalg.norm(want[0])
    # updated operators equal operators created from the same blocks (set_blocks is a pure re-fill)
    blocks = [np.asfortranarray(b.cpu().numpy()) for b in setup.blocks(Sd)]
    S2 = api.LocalSchurs(ctx, blocks, sub.gather_idx, sub.node_Γ_cnt)
    assert np.array_equal(S2 * v, ys)


def test_full_size_device_setup(pkg, ctx, fem):
    """Config 3 (1 M DoF, 4x2): S_d and ΠS_d of all eight subdomains on the device, against the host blocks; prints the
    per-realization set-up time."""
    import time
    import torch
    api = pkg.api
    mesh = fem.get_mesh(1000)
    P = fem.build_schur_problem(1000, 4, 2, lognormal_coeff(fem, mesh.points), f_m1, u0734)
    sub = P.sub
    t0 = time.perf_counter()
    setup = api.SchurSetup(ctx, P.A_IIdd, P.A_IΓdd, P.A_ΓΓdd)
    t_plan = time.perf_counter() - t0
    vals = [torch.from_numpy(v).cuda() for v in setup._vals]
    bI = torch.from_numpy(np.concatenate(P.b_Id)).cuda()
    setup.run(*vals, bI); ctx.synchronize()                                   # warm-up (library handles, work space)
    t0 = time.perf_counter()
    Sd, w = setup.run(*vals, bI)
    ctx.synchronize()
    t_S = time.perf_counter() - t0
    t0 = time.perf_counter()
    Pi = api.nn_pinv(ctx, sub.n_Γd, Sd)
    ctx.synchronize()
    t_pinv = time.perf_counter() - t0
    print(f"\\nconfig 3 set-up on the device: plan {t_plan:.2f} s (once), S_d + w_d {t_S * 1e3:.1f} ms, pinv {t_pinv * 1e3:.1f} ms per realization")
    blocks = [b.cpu().numpy() for b in setup.blocks(Sd)]
    for d in range(sub.ndom):
        assert np.abs(blocks[d] - P.Sd[d]).max() <= 1e-9 * np.abs(P.Sd[d]).max()
    Pib = [b.cpu().numpy() for b in setup.blocks(Pi)]
    for d in (0, 1):
        assert np.abs(Pib[d] - P.ΠSd[d]).max() <= 1e-6 * np.abs(P.ΠSd[d]).max()


@pytest.mark.parametrize("N,px,py,seed", [(50, 3, 2, 7), (90, 4, 2, 5)])
def test_level_solves_exact_interior_solve_on_the_device(pkg, ctx, orc, fem, N, px, py, seed):
    """`mi_schur_setup_keep_levels` / `mi_schur_setup_interior_solve` / `mi_schur_matfree_interior_levels`: the interior solve
    of the matrix-free applies (EPDD.jl:648-650: `IterativeSolvers.cg(A_IIdd, A_IΓdd xd; reltol)`) done EXACTLY on the device by
    sweeps over the level inverses the set-up keeps. Against sparse direct solves on the host (SuperLU), the oracle's
    matrix-free apply with those solves, the assembled operator (Example03:175), and `get_schur_rhs` /
    `get_subdomain_solutions` through the same operator."""
    api = pkg.api
    mesh = fem.get_mesh(N)
    P = fem.build_schur_problem(N, px, py, lognormal_coeff(fem, mesh.points, seed), f_m1, u0734)
    sub = P.sub
    n = sub.n_Γ
    setup = api.SchurSetup(ctx, P.A_IIdd, P.A_IΓdd, P.A_ΓΓdd)
    setup.keep_levels(True)
    Sd, _ = setup.run(b_I=np.concatenate(P.b_Id))
    for d, blk in enumerate(setup.blocks(Sd)):                          # keeping the levels does not change the run
        assert np.abs(blk - P.Sd[d]).max() <= 1e-10 * np.abs(P.Sd[d]).max()
    rng = np.random.default_rng(1)
    f = rng.standard_normal(int(sum(setup.n_Id)))
    u = setup.interior_solve(f)
    off = 0
    for d in range(sub.ndom):
        ni = setup.n_Id[d]
        ref = P.solvers[d](f[off:off + ni])
        assert np.abs(u[off:off + ni] - ref).max() <= 1e-10 * np.abs(ref).max(), d
        off += ni
    # the matrix-free operator with these solves: equals the oracle's with sparse direct solves, and the assembled operator
    Smf = api.MatrixFreeLocalSchurs(ctx, P.A_IIdd, P.A_IΓdd, P.A_ΓΓdd, sub.gather_idx, sub.node_Γ_cnt, None, reltol=1e-9)
    Smf.use_level_solver(setup)
    So_mf = orc.apply_local_schurs_matfree_operator(P.A_IIdd, P.A_IΓdd, P.A_ΓΓdd, sub.gather_idx, n, P.solvers)
    So = orc.apply_local_schurs_operator(P.Sd, sub.gather_idx, n)
    v = rng.standard_normal(n)
    y = Smf * v
    assert np.abs(y - So_mf * v).max() <= 1e-10 * np.abs(y).max()
    assert np.abs(y - So * v).max() <= 1e-10 * np.abs(y).max()          # Example03:175 with an exact interior solve
    b_schur = Smf.schur_rhs(np.concatenate(P.b_Id), P.b_Γ)              # get_schur_rhs, EPDD.jl:835-864
    assert np.abs(b_schur - P.b_schur).max() <= 1e-10 * np.abs(P.b_schur).max()
    # a Schur solve with the matrix-free operator, then the interiors (Example03:196-204)
    M = api.NeumannNeumannSchurPreconditioner(ctx, P.ΠSd, sub.gather_idx, sub.node_Γ_cnt)
    x, it, res = api.pcg(Smf, b_schur, np.zeros(n), M)
    want = orc.pcg(So, P.b_schur, np.zeros(n), orc.neumann_neumann_operator(P.ΠSd, sub.gather_idx, sub.node_Γ_cnt))
    assert it == want[1] and np.linalg.norm(x - want[0]) <= 1e-6 * np.linalg.norm(want[0])
    Smf.use_level_solver(None)                                          # back to the interior CG of the operator
    y2 = Smf * v
    assert np.abs(y2 - y).max() <= 1e-7 * np.abs(y).max()
