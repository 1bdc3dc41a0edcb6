"""GPU suite (-m gpu), SURVEY.md §8 row f1: eigcg / eigpcg / eigdefcg / eigdefpcg / initcg / initpcg through the C ABI
against the numpy restatement in oracle/oracle.py (eigcg.jl, defcg.jl:111-223, 337-473, initcg.jl).

What is compared
  * (x, it, res_norm): the bar of the cg/pcg/defcg/defpcg tests (`assert_history`) — the Krylov part of these
    solvers is untouched by the Ritz extraction (eigdefpcg adds the re-orthogonalisation of r against W).
  * V[:, 1:nvec]: eigenvectors are defined up to sign (and rotation inside clusters) and the reference's LAPACK
    calls are restated with Jacobi solvers on the host, so the SUBSPACES are compared: largest principal angle
    sin θ <= SUBSPACE_TOL, and the Ritz values (generalised Rayleigh quotients) to RITZ_RTOL.
PARITY UNPINNED (no reference fixture exists for these either; see oracle/krylov_oracle.c).
"""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import a_example01, f_m1, u3
from test_gpu_parity import assert_history, gpu_ops, orc_ops

pytestmark = pytest.mark.gpu

SUBSPACE_TOL = 1e-6
RITZ_RTOL = 1e-6      # the Lanczos vectors are iterates (z/sqrt(rTz)): they carry the X_RTOL-level differences of the solve


def sin_theta(V1, V2):
    Q1, _ = np.linalg.qr(V1)
    Q2, _ = np.linalg.qr(V2)
    return np.linalg.norm(Q2 - Q1 @ (Q1.T @ Q2), 2)


def ritz_values(apply, V):
    """Eigenvalues of (V'AV, V'V): what the recycled space 'sees' of A."""
    AV = np.column_stack([apply(V[:, j]) for j in range(V.shape[1])])
    import scipy.linalg as sla
    return sla.eigh(V.T @ AV, V.T @ V, eigvals_only=True)


def assert_space(apply, V, Vo, tol=SUBSPACE_TOL):
    assert V.shape == Vo.shape and np.all(np.isfinite(V))
    s = sin_theta(V, Vo)
    assert s <= tol, f"sin(theta_max) = {s:.3e}"
    assert np.allclose(ritz_values(apply, V), ritz_values(apply, Vo), rtol=RITZ_RTOL)


@pytest.fixture(scope="module")
def fullA(fem):
    mesh = fem.get_mesh(40)
    d = fem.get_dirichlet_inds(mesh.points, mesh.point_marker)
    A, b = fem.do_isotropic_elliptic_assembly(mesh.cells, mesh.points, d, mesh.point_marker, a_example01, f_m1, u3)
    return sp.csr_matrix(A), b


@pytest.mark.parametrize("nvec,spdim", [(6, 20), (3, 7), (10, 24)])
def test_eigcg_full_matrix(pkg, ctx, orc, fullA, nvec, spdim):
    A, b = fullA
    n = A.shape[0]
    Ao = orc.csc_operator(A)
    want = orc.eigcg(Ao, b, np.zeros(n), nvec, spdim)
    got = pkg.api.eigcg(pkg.api.SparseMatrixCSC(ctx, A), b, np.zeros(n), nvec, spdim)
    assert want[1] > spdim + 2 * (spdim - 2 * nvec), "case must go through several restarts"
    assert_history(got[:3], want[:3], apply=Ao, b=b)
    assert_space(Ao, got[3], want[3])


def test_eigpcg_jacobi_full_matrix(pkg, ctx, orc, fullA):
    A, b = fullA
    n = A.shape[0]
    Ao, Mo = orc.csc_operator(A), orc.jacobi_operator(A.diagonal())
    want = orc.eigpcg(Ao, b, np.zeros(n), Mo, 6, 20)
    api = pkg.api
    got = api.eigpcg(api.SparseMatrixCSC(ctx, A), b, np.zeros(n), api.JacobiPreconditioner(ctx, A.diagonal()), 6, 20)
    assert_history(got[:3], want[:3], apply=Ao, b=b)
    assert_space(Ao, got[3], want[3])


def test_eigcg_on_schur_operator(pkg, ctx, orc, toy):
    S, _ = gpu_ops(pkg, ctx, toy)
    So, _ = orc_ops(orc, toy)
    n = toy.sub.n_Γ
    want = orc.eigcg(So, toy.b_schur, np.zeros(n), 5, 14)
    got = pkg.api.eigcg(S, toy.b_schur, np.zeros(n), 5, 14)
    assert_history(got[:3], want[:3], apply=So, b=toy.b_schur)
    assert_space(So, got[3], want[3])


@pytest.mark.parametrize("case", ["toy", "ragged"])
def test_eigpcg_then_eigdefpcg_nn_schur(pkg, ctx, orc, toy, ragged, case):
    """Example09's recycling chain on the NN-preconditioned Schur system: eigpcg on the first system, eigdefpcg
    (W = the previous V) on the next right-hand side; nvec = floor(1.25 ndom), spdim = 3 ndom (Example09:39-40)."""
    P = toy if case == "toy" else ragged
    ndom = len(P.Sd)
    nvec, spdim = (int(1.25 * ndom), 3 * ndom) if case == "ragged" else (3, 8)   # toy: see test_final_extraction_bounds_error
    S, M = gpu_ops(pkg, ctx, P)
    So, Mo = orc_ops(orc, P)
    n = P.sub.n_Γ
    api = pkg.api
    want = orc.eigpcg(So, P.b_schur, np.zeros(n), Mo, nvec, spdim)
    got = api.eigpcg(S, P.b_schur, np.zeros(n), M, nvec, spdim)
    assert_history(got[:3], want[:3])
    assert_space(So, got[3], want[3])
    # next system: same operator, new right-hand side; both sides deflate with the ORACLE's W so that the
    # comparison is of the solver, not of the previous step's rounding
    b2 = So(np.random.default_rng(3).standard_normal(n))
    W = want[3]
    want2 = orc.eigdefpcg(So, b2, np.zeros(n), Mo, W, spdim)
    got2 = api.eigdefpcg(S, b2, np.zeros(n), M, W, spdim)
    assert_history(got2[:3], want2[:3])
    assert_space(So, got2[3], want2[3])
    assert got2[1] <= orc.pcg(So, b2, np.zeros(n), Mo)[1]          # deflation never costs iterations here
    # and the chain run end to end on the device (W from the device's own eigpcg) converges in as many iterations
    got3 = api.eigdefpcg(S, b2, np.zeros(n), M, got[3], spdim)
    assert abs(got3[1] - want2[1]) <= 1


def test_eigdefcg_and_eigdefpcg_full_matrix(pkg, ctx, orc, fullA):
    A, b = fullA
    n = A.shape[0]
    api = pkg.api
    Ao, Mo = orc.csc_operator(A), orc.jacobi_operator(A.diagonal())
    Ag, Mg = api.SparseMatrixCSC(ctx, A), api.JacobiPreconditioner(ctx, A.diagonal())
    b2 = A @ np.random.default_rng(11).standard_normal(n)
    W = orc.eigcg(Ao, b, np.zeros(n), 6, 20)[3]
    want = orc.eigdefcg(Ao, b2, np.zeros(n), W, 20)
    got = api.eigdefcg(Ag, b2, np.zeros(n), W, 20)
    assert_history(got[:3], want[:3], apply=Ao, b=b2)
    assert_space(Ao, got[3], want[3])
    Wp = orc.eigpcg(Ao, b, np.zeros(n), Mo, 6, 20)[3]
    want = orc.eigdefpcg(Ao, b2, np.zeros(n), Mo, Wp, 20)
    got = api.eigdefpcg(Ag, b2, np.zeros(n), Mg, Wp, 20)
    assert_history(got[:3], want[:3], apply=Ao, b=b2)
    assert_space(Ao, got[3], want[3])


def test_initcg_initpcg(pkg, ctx, orc, toy):
    S, M = gpu_ops(pkg, ctx, toy)
    So, Mo = orc_ops(orc, toy)
    n = toy.sub.n_Γ
    api = pkg.api
    W = orc.eigpcg(So, toy.b_schur, np.zeros(n), Mo, 3, 8)[3]
    b2 = So(np.random.default_rng(4).standard_normal(n))
    assert_history(api.initpcg(S, b2, np.zeros(n), M, W), orc.initpcg(So, b2, np.zeros(n), Mo, W))
    assert_history(api.initcg(S, b2, np.zeros(n), W), orc.initcg(So, b2, np.zeros(n), W), apply=So, b=b2)


def test_eig_device_pointers_and_chunking(pkg, ctx, orc, toy):
    """torch CUDA tensors in, column-major V out on the device; eager launches (chunk 0) give the same bits."""
    import torch
    S, M = gpu_ops(pkg, ctx, toy)
    n = toy.sub.n_Γ
    api = pkg.api
    ref = api.eigpcg(S, toy.b_schur, np.zeros(n), M, 3, 8)
    b = torch.from_numpy(toy.b_schur).cuda()
    x = torch.zeros(n, dtype=torch.float64, device="cuda")
    xd, it, res, V = api.eigpcg(S, b, x, M, 3, 8)
    assert it == ref[1] and np.array_equal(res, ref[2]) and np.array_equal(x.cpu().numpy(), ref[0])
    assert V.is_cuda and np.array_equal(V.cpu().numpy(), ref[3])
    ctx.set_chunk(0)
    try:
        eager = api.eigpcg(S, toy.b_schur, np.zeros(n), M, 3, 8)
    finally:
        ctx.set_chunk(8)
    assert eager[1] == ref[1] and np.array_equal(eager[2], ref[2]) and np.array_equal(eager[3], ref[3])


def test_final_extraction_bounds_error(pkg, ctx, orc, toy):
    """Example09's sizes on the 4-subdomain toy (nvec 5, spdim 12): NN-PCG stops after 5 loop iterations, the final
    Ritz extraction then asks for eigvecs(Tm[1:4,1:4])[:, 1:5] — a BoundsError in the reference (eigcg.jl:275), in the
    oracle and (MI_ERR_BOUNDS) on the device."""
    S, M = gpu_ops(pkg, ctx, toy)
    So, Mo = orc_ops(orc, toy)
    n = toy.sub.n_Γ
    with pytest.raises(orc.BoundsError):
        orc.eigpcg(So, toy.b_schur, np.zeros(n), Mo, 5, 12)
    with pytest.raises(pkg.api.BoundsError):
        pkg.api.eigpcg(S, toy.b_schur, np.zeros(n), M, 5, 12)


def test_eig_errors(pkg, ctx, toy):
    S, M = gpu_ops(pkg, ctx, toy)
    n = toy.sub.n_Γ
    api = pkg.api
    with pytest.raises(api.BoundsError):
        api.eigpcg(S, toy.b_schur, np.zeros(n), M, 5, 10)           # spdim < 2 nvec + 1
    with pytest.raises(api.MiError):
        api.eigcg(S, toy.b_schur, np.zeros(n), 0, 10)
    W = np.zeros((n, 3))
    with pytest.raises(api.SingularException):
        api.eigdefpcg(S, toy.b_schur, np.zeros(n), M, W, 9)         # WtAW singular, as in defpcg


def test_eig_multi_workgroup_loop_matches_fused(pkg, ctx, orc, toy, monkeypatch):
    """Large systems (n > 8192) run the eigCG family on the multi-workgroup loop kernels; force that path on the toy
    system and compare with the fused-loop result and the oracle."""
    S, M = gpu_ops(pkg, ctx, toy)
    So, Mo = orc_ops(orc, toy)
    n = toy.sub.n_Γ
    api = pkg.api
    fused = api.eigpcg(S, toy.b_schur, np.zeros(n), M, 3, 8)
    fused_cg = api.eigcg(S, toy.b_schur, np.zeros(n), 5, 14)
    monkeypatch.setenv("MI355_NO_FUSED", "1")
    got = api.eigpcg(S, toy.b_schur, np.zeros(n), M, 3, 8)
    got_cg = api.eigcg(S, toy.b_schur, np.zeros(n), 5, 14)
    monkeypatch.delenv("MI355_NO_FUSED")
    assert_history(got[:3], orc.eigpcg(So, toy.b_schur, np.zeros(n), Mo, 3, 8)[:3])
    assert got[1] == fused[1] and np.allclose(got[2], fused[2], rtol=1e-8, atol=1e-12 * fused[2][0])
    assert_space(So, got[3], fused[3])
    assert_history(got_cg[:3], fused_cg[:3], apply=So, b=toy.b_schur)
    assert_space(So, got_cg[3], fused_cg[3])


def test_eigpcg_large_system_multi_workgroup(pkg, ctx, orc, fem):
    """n = 9 604 > 8 192: the genuinely multi-workgroup path (CSR SpMV with the dot in its epilogue, Jacobi folded into the
    r-update, recording kernels over several workgroups). A long Jacobi-PCG run: history to the long-run bar, the recycled
    space to 1e-3 (its vectors are late CG iterates, which drift between summation orders — see assert_history)."""
    mesh = fem.get_mesh(100)
    d = fem.get_dirichlet_inds(mesh.points, mesh.point_marker)
    A, b = fem.do_isotropic_elliptic_assembly(mesh.cells, mesh.points, d, mesh.point_marker, a_example01, f_m1, u3)
    A = sp.csr_matrix(A)
    n = A.shape[0]
    assert n > 8192
    api = pkg.api
    Ao, Mo = orc.csc_operator(A), orc.jacobi_operator(A.diagonal())
    want = orc.eigpcg(Ao, b, np.zeros(n), Mo, 6, 20)
    got = api.eigpcg(api.SparseMatrixCSC(ctx, A), b, np.zeros(n), api.JacobiPreconditioner(ctx, A.diagonal()), 6, 20)
    assert_history(got[:3], want[:3], apply=Ao, b=b)
    assert np.all(np.isfinite(got[3])) and sin_theta(got[3], want[3]) <= 1e-3
    assert np.allclose(ritz_values(Ao, got[3]), ritz_values(Ao, want[3]), rtol=1e-3)
