"""GPU suite (-m gpu), SURVEY.md §8 row f3: the numeric half of `prepare_local_schurs` (EPDD.jl:389-546) on the device,
per-realization value updates of the matrix-free operator and `get_schur_rhs` (EPDD.jl:835-864) with the device
interior solve.

Bar: the assembled values are BIT-EXACT against the host element loop (same operations in the same order,
-ffp-contract=off) and against the numpy plan executor of the oracle; an operator updated in place applies
bit-identically to one created from the same values; the condensed right-hand side agrees with sparse-direct
interior solves to the interior CG's tolerance (reltol 1e-9 on the interior residual)."""
import numpy as np
import pytest

from conftest import f_m1, lognormal_coeff, u0734

pytestmark = pytest.mark.gpu


def _setup(fem, N, px, py, f, ue):
    mesh = fem.get_mesh(N)
    dinds = fem.get_dirichlet_inds(mesh.points, mesh.point_marker)
    epart, npart = fem.mesh_partition(mesh, px, py)
    sub = fem.set_subdomains(mesh.cells, mesh.cell_neighbors, epart, npart, dinds.dirichlet_g2l)
    return mesh, epart, sub, fem.make_assembly_plan(mesh.cells, mesh.points, epart, sub, f, ue)


@pytest.mark.parametrize("N,px,py", [(40, 2, 2), (50, 3, 2), (100, 4, 2)])
def test_device_assembly_is_bit_exact(pkg, ctx, orc, fem, N, px, py):
    f = lambda x, y: np.sin(3 * x) + y          # noqa: E731
    ue = lambda x, y: 0.5 + x * y               # noqa: E731
    mesh, epart, sub, plan = _setup(fem, N, px, py, f, ue)
    dev = pkg.api.AssemblyPlan(ctx, plan)
    for seed in (7, 8):
        a = lognormal_coeff(fem, mesh.points, seed)
        vals = dev.run(a)
        assert np.array_equal(vals, orc.run_assembly_plan(plan, a))
        got = plan.blocks(vals)
        want = fem.prepare_local_schurs(mesh.cells, mesh.points, epart, sub, a, f, ue)
        for k in range(3):
            for A, B in zip(got[k], want[k]):
                assert np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices)
                assert np.array_equal(A.data, B.data)
        assert all(np.array_equal(u, v) for u, v in zip(got[3], want[3])) and np.array_equal(got[4], want[4])
    import torch
    vd = dev.run(torch.from_numpy(a).cuda())
    ctx.synchronize()                                    # device-pointer calls are asynchronous on the context's stream
    assert vd.is_cuda and np.array_equal(vd.cpu().numpy(), vals)


def test_realization_update_on_the_device(pkg, ctx, orc, fem):
    """Example07's per-realization flow without the host: assemble -> set_values -> schur_rhs -> pcg, against the host flow."""
    import torch
    api = pkg.api
    N, px, py = 60, 3, 2
    mesh, epart, sub, plan = _setup(fem, N, px, py, f_m1, u0734)
    a0, a1 = lognormal_coeff(fem, mesh.points, 1), lognormal_coeff(fem, mesh.points, 2)
    P0 = fem.build_schur_problem(N, px, py, a0, f_m1, u0734)
    P1 = fem.build_schur_problem(N, px, py, a1, f_m1, u0734)
    gi, cnt = sub.gather_idx, sub.node_Γ_cnt
    S = api.MatrixFreeLocalSchurs(ctx, P0.A_IIdd, P0.A_IΓdd, P0.A_ΓΓdd, gi, cnt, None, reltol=1e-11)
    dev = api.AssemblyPlan(ctx, plan)
    vals = dev.run(torch.from_numpy(a1).cuda())                       # stays on the device
    ii, ig, gg, bI, bΓ = dev.block_values(vals)
    S.set_values(ii, ig, gg)
    S_fresh = api.MatrixFreeLocalSchurs(ctx, P1.A_IIdd, P1.A_IΓdd, P1.A_ΓΓdd, gi, cnt, None, reltol=1e-11)
    x = np.random.default_rng(0).standard_normal(sub.n_Γ)
    assert np.array_equal(S * x, S_fresh * x)                          # same values, same kernels: same bits
    b_dev = S.schur_rhs(bI, bΓ).cpu().numpy()
    assert np.linalg.norm(b_dev - P1.b_schur) <= 1e-8 * np.linalg.norm(P1.b_schur)
    # back-substitution on the device (get_subdomain_solutions, EPDD.jl:1014-1025) against sparse-direct interior solves
    u_I = S.interior_solutions(torch.from_numpy(x).cuda(), bI).cpu().numpy()
    ref = np.concatenate([P1.solvers[d](P1.b_Id[d] - P1.A_IΓdd[d] @ x[gi[d]]) for d in range(sub.ndom)])
    assert np.linalg.norm(u_I - ref) <= 1e-8 * np.linalg.norm(ref)
    # the update of one block only: others keep their values
    S.set_values(None, None, gg)
    assert np.array_equal(S * x, S_fresh * x)
    # solve the new realization with the ξ-fixed preconditioner (Example07:273) on both paths
    M0 = api.NeumannNeumannSchurPreconditioner(ctx, P0.ΠSd, gi, cnt)
    got = api.pcg(S, b_dev, np.zeros(sub.n_Γ), M0)
    want = api.pcg(api.LocalSchurs(ctx, P1.Sd, gi, cnt), P1.b_schur, np.zeros(sub.n_Γ), M0)
    assert got[1] == want[1]
    assert np.linalg.norm(got[0] - want[0]) <= 1e-6 * np.linalg.norm(want[0])
    # host-pointer mode and the host-callback operator (A_II stays with the callback)
    Sh = api.MatrixFreeLocalSchurs(ctx, P0.A_IIdd, P0.A_IΓdd, P0.A_ΓΓdd, gi, cnt, P1.solvers)
    vh = dev.run(a1)
    iih, igh, ggh, bIh, bΓh = dev.block_values(vh)
    Sh.set_values(None, igh, ggh)
    assert np.allclose(Sh * x, S_fresh * x, rtol=1e-8, atol=1e-10 * np.abs(x).max())
    assert np.linalg.norm(Sh.schur_rhs(bIh, bΓh) - P1.b_schur) <= 1e-10 * np.linalg.norm(P1.b_schur)
    with pytest.raises(api.MiError):
        Sh.set_values(iih, None, None)


def test_assembly_plan_errors(pkg, ctx, fem):
    import copy
    _, _, _, plan = _setup(fem, 20, 2, 2, f_m1, u0734)
    bad = copy.copy(plan)
    bad.ccode = plan.ccode.copy()
    bad.ccode[0] = 12 * plan.cells.shape[1]                            # element index out of range
    with pytest.raises(pkg.api.MiError):
        pkg.api.AssemblyPlan(ctx, bad)
    S = pkg.api.LocalSchurs(ctx, [np.eye(2)], [np.arange(2)], np.ones(2, dtype=np.int64))
    with pytest.raises(pkg.api.MiError):
        pkg.api.MatrixFreeLocalSchurs.set_values(S, None, None, None)  # not a matrix-free operator
