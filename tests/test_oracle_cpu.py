"""CPU suite (-m "not gpu"): pins the oracle and the host-side set-up.

The reference ships no tests or golden vectors for this path, so the oracle is
pinned by (1) agreement of the C restatement with independent numpy expressions, (2) the identities
the reference prints at run time (Example03:175 and :204), (3) known-answer tests, and (4) the
committed fixtures in tests/golden/ (made by tests/golden/make_golden.py from this same oracle —
a regression pin, not an external one: PARITY UNPINNED against Julia itself).
"""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from conftest import GOLDEN, a_example01, f_m1, lowest_eigvecs, one, u0734, u3


# ------------------------------------------------------------------ C restatement vs numpy
def test_blas1_and_spmv_match_numpy(orc, toy):
    rng = np.random.default_rng(0)
    A = sp.csc_matrix(toy.A_IIdd[0])
    x = rng.standard_normal(A.shape[0])
    op = orc.csc_operator(A)
    opg = orc.csc_operator(A, gather=True)
    y = op(x)
    assert np.array_equal(y, opg(x))            # scatter (CSC) and gather (CSR) orders agree on symmetric A
    assert np.allclose(y, A @ x, rtol=1e-13, atol=1e-13)
    # left-to-right row sums, no FMA: reproduce with a python loop on a few rows
    Ar = sp.csr_matrix(A)
    for i in (0, 17, A.shape[0] - 1):
        s = 0.0
        for k in range(Ar.indptr[i], Ar.indptr[i + 1]):
            s += Ar.data[k] * x[Ar.indices[k]]
        assert s == y[i]


def test_lu_solve_matches_numpy(orc):
    rng = np.random.default_rng(1)
    A = rng.standard_normal((14, 14))
    b = rng.standard_normal(14)
    assert np.allclose(orc.lu_solve(A, b), np.linalg.solve(A, b), rtol=1e-10)
    with pytest.raises(orc.SingularException):
        orc.lu_solve(np.zeros((3, 3)), np.ones(3))


def test_assembled_and_nn_apply_match_numpy(orc, ragged):
    P = ragged
    n = P.sub.n_Γ
    rng = np.random.default_rng(2)
    x = rng.standard_normal(n)
    S = orc.apply_local_schurs_operator(P.Sd, P.sub.gather_idx, n)
    M = orc.neumann_neumann_operator(P.ΠSd, P.sub.gather_idx, P.sub.node_Γ_cnt)
    ys, ym = np.zeros(n), np.zeros(n)
    cnt = P.sub.node_Γ_cnt.astype(float)
    for d in range(P.sub.ndom):
        g = P.sub.gather_idx[d]
        ys[g] += P.Sd[d] @ x[g]
        ym[g] += (P.ΠSd[d] @ (x[g] / cnt[g])) / cnt[g]
    assert np.allclose(S(x), ys, rtol=1e-12, atol=1e-12)
    assert np.allclose(M(x), ym, rtol=1e-12, atol=1e-12)


# ------------------------------------------------------------------ set-up facts (SURVEY.md §8 sizes)
def test_toy_sizes_and_maps(toy):
    s = toy.sub
    assert s.n_Γ == 195 and sorted(s.n_Γd) == [97, 98, 98, 99]
    assert sum(s.n_Id) + s.n_Γ == 98 * 98 == 9604
    assert set(np.unique(s.node_Γ_cnt)) == {2, 4} and (s.node_Γ_cnt == 4).sum() == 1
    for d in range(s.ndom):
        assert np.array_equal(s.node_Γ[s.gather_idx[d]], s.node_Γd[d])
        assert len(set(s.gather_idx[d])) == s.n_Γd[d]
    assert np.array_equal(np.bincount(np.concatenate(s.gather_idx), minlength=s.n_Γ), s.node_Γ_cnt)


def test_blocks_are_symmetric_and_consistent_with_full_matrix(fem, toy):
    P = toy
    A, b = fem.do_isotropic_elliptic_assembly(P.mesh.cells, P.mesh.points, P.dinds, P.mesh.point_marker, one, f_m1, u0734)
    assert abs(A - A.T).max() == 0.0           # bitwise symmetric (EPDD.jl:294 ΔKij is symmetric in i,j)
    for d in range(P.sub.ndom):
        assert abs(P.A_IIdd[d] - P.A_IIdd[d].T).max() == 0.0
        assert abs(P.A_ΓΓdd[d] - P.A_ΓΓdd[d].T).max() == 0.0
        # interior block of the full matrix == A_II of the subdomain
        rows = P.dinds.not_dirichlet_g2l[P.sub.node_Id[d]]
        assert abs(A[rows][:, rows] - P.A_IIdd[d]).max() < 1e-14


# ------------------------------------------------------------------ identities the reference prints
def test_three_schur_formulations_agree(fem, orc, ragged):
    """Example03:175 `extrema(S_global*b_schur - S_local_mat*b_schur)`; exact to rounding with a direct interior solve."""
    P = ragged
    mesh = P.mesh
    coeff = P.info.get("coeff")
    n = P.sub.n_Γ
    Sa = orc.apply_local_schurs_operator(P.Sd, P.sub.gather_idx, n)
    Sm = orc.apply_local_schurs_matfree_operator(P.A_IIdd, P.A_IΓdd, P.A_ΓΓdd, P.sub.gather_idx, n, P.solvers)
    v = P.b_schur
    assert np.allclose(Sa(v), Sm(v), rtol=1e-11, atol=1e-12 * np.abs(Sa(v)).max())


def test_schur_solution_matches_direct_solve(fem, orc, toy):
    """Example03:193-204: NN-PCG on S, back-substitute, merge; compare with the full-system solve."""
    P = toy
    n = P.sub.n_Γ
    S = orc.apply_local_schurs_operator(P.Sd, P.sub.gather_idx, n)
    M = orc.neumann_neumann_operator(P.ΠSd, P.sub.gather_idx, P.sub.node_Γ_cnt)
    uΓ, it, res = orc.pcg(S, P.b_schur, np.zeros(n), M)
    assert it < 20 and res[-1] <= 1e-7 * np.linalg.norm(P.b_schur)
    A_IIg, A_IΓg, A_ΓΓ, b_Id, b_Γ = fem.prepare_global_schur(P.mesh.cells, P.mesh.points, P.epart, P.sub, one, f_m1, u0734)
    Sg = orc.apply_global_schur_operator(A_IIg, A_IΓg, A_ΓΓ, P.solvers)
    assert np.allclose(Sg(P.b_schur), S(P.b_schur), rtol=1e-11, atol=1e-13)
    u_dd = fem.merge_subdomain_solutions(uΓ, fem.get_subdomain_solutions(uΓ, A_IIg, A_IΓg, b_Id, P.solvers),
                                         P.sub, P.dinds, u0734, P.mesh.points)
    A, b = fem.do_isotropic_elliptic_assembly(P.mesh.cells, P.mesh.points, P.dinds, P.mesh.point_marker, one, f_m1, u0734)
    u = fem.append_bc(P.dinds, spla.spsolve(sp.csc_matrix(A), b), P.mesh.points, u0734)
    assert np.abs(u_dd - u).max() < 1e-8


# ------------------------------------------------------------------ known answers
def test_constant_and_affine_solutions_are_exact(fem, orc):
    """P1 reproduces affine functions: f = 0, u = uexact on the boundary => nodal solution exact."""
    zero = lambda x, y: 0.0 * x
    for uex in (u0734, lambda x, y: 2.0 * x - 3.0 * y + 0.5):
        P = fem.build_schur_problem(30, 3, 2, one, zero, uex)
        n = P.sub.n_Γ
        S = orc.apply_local_schurs_operator(P.Sd, P.sub.gather_idx, n)
        M = orc.neumann_neumann_operator(P.ΠSd, P.sub.gather_idx, P.sub.node_Γ_cnt)
        uΓ, it, _ = orc.pcg(S, P.b_schur, np.zeros(n), M)
        pts = P.mesh.points[:, P.sub.node_Γ]
        assert np.abs(uΓ - uex(pts[0], pts[1])).max() < 1e-6


def test_floating_subdomain_schur_is_singular_with_constant_kernel(fem):
    """S_d * 1 = 0 on a subdomain that touches no Dirichlet node; pinv drops exactly that mode."""
    P = fem.build_schur_problem(31, 3, 3, one, f_m1, u0734)
    d = 4                                                  # the centre box
    assert P.dinds.dirichlet_g2l[P.mesh.cells[:, P.sub.elemd[d]]].max() == -1
    S = P.Sd[d]
    assert np.abs(S @ np.ones(S.shape[0])).max() < 1e-10 * np.abs(S).max()
    assert np.linalg.matrix_rank(P.ΠSd[d], tol=1e-8) == S.shape[0] - 1
    assert np.abs(S - S.T).max() == 0.0                    # `Symmetric(...)`, EPDD.jl:692


def test_operator_symmetry_and_nn_semidefinite(orc, ragged):
    P = ragged
    n = P.sub.n_Γ
    rng = np.random.default_rng(3)
    v, w = rng.standard_normal(n), rng.standard_normal(n)
    S = orc.apply_local_schurs_operator(P.Sd, P.sub.gather_idx, n)
    M = orc.neumann_neumann_operator(P.ΠSd, P.sub.gather_idx, P.sub.node_Γ_cnt)
    assert abs(v @ S(w) - w @ S(v)) < 1e-10 * abs(v @ S(w))
    assert abs(v @ M(w) - w @ M(v)) < 1e-8 * max(1.0, abs(v @ M(w)))
    assert v @ S(v) > 0 and v @ M(v) >= -1e-12


def test_cg_semantics(orc, micro):
    P = micro
    n = P.sub.n_Γ
    S = orc.apply_local_schurs_operator(P.Sd, P.sub.gather_idx, n)
    b = P.b_schur
    x, it, res = orc.cg(S, b, np.zeros(n))
    assert it <= n and res.size == it                      # SPD n x n: at most n iterations; res_norm[1:it]
    assert res[0] == np.sqrt(np.dot(b, b)) or np.isclose(res[0], np.linalg.norm(b), rtol=1e-15)
    assert res[-1] <= 1e-7 * np.linalg.norm(b) < res[-2]   # stops at the first entry under tol
    # maxit: `while it < maxit` => exactly maxit entries, maxit-1 loop passes (cg.jl:34)
    x3, it3, res3 = orc.cg(S, b, np.zeros(n), maxit=3)
    assert it3 == 3 and np.array_equal(res3, res[:3])
    # exact initial guess: it = 1, no loop pass
    x1, it1, res1 = orc.cg(S, b, x)
    assert it1 == 1 and res1[0] <= 1e-7 * np.linalg.norm(b)


def test_deflation_semantics(orc, toy):
    P = toy
    n = P.sub.n_Γ
    S = orc.apply_local_schurs_operator(P.Sd, P.sub.gather_idx, n)
    M = orc.neumann_neumann_operator(P.ΠSd, P.sub.gather_idx, P.sub.node_Γ_cnt)
    b, x0 = P.b_schur, np.zeros(n)
    xp, itp, resp = orc.pcg(S, b, x0, M)
    # W with zero columns: defpcg == pcg, defcg == cg, entry for entry
    W0 = np.zeros((n, 0), order="F")
    xd, itd, resd = orc.defpcg(S, b, x0, W0, M)
    assert itd == itp and np.array_equal(resd, resp) and np.array_equal(xd, xp)
    xc, itc, resc = orc.cg(S, b, x0)
    xdc, itdc, resdc = orc.defcg(S, b, x0, W0)
    assert itdc == itc and np.array_equal(resdc, resc)
    # deflating the nev lowest eigenvectors (Example03:206-214, nev = ndom + 10) cannot slow CG down
    W = lowest_eigvecs(S, n, P.sub.ndom + 10)
    xw, itw, _ = orc.defcg(S, b, x0, W)
    assert itw < itc and np.linalg.norm(xw - xc) < 1e-5 * np.linalg.norm(xc)
    xq, itq, _ = orc.defpcg(S, b, x0, W, M)
    assert itq <= itp and np.linalg.norm(xq - xp) < 1e-5 * np.linalg.norm(xp)
    # rank-deficient W => SingularException from `WtAW \ mu` (README "To do")
    Wbad = np.asfortranarray(np.column_stack([W[:, 0], np.zeros(n)]))   # an exactly zero pivot for any value of w'Sw
    with pytest.raises(orc.SingularException):
        orc.defpcg(S, b, x0, Wbad, M)


def test_full_system_pcg_config2_shape(fem, orc):
    """Example01 flow at a small N: full A, pcg with Jacobi / none standing in for AMG."""
    mesh = fem.get_mesh(40)
    d = fem.get_dirichlet_inds(mesh.points, mesh.point_marker)
    A, b = fem.do_isotropic_elliptic_assembly(mesh.cells, mesh.points, d, mesh.point_marker, a_example01, f_m1, u3)
    Aop = orc.csc_operator(A)
    x, it, res = orc.pcg(Aop, b, np.zeros(b.size), orc.jacobi_operator(A.diagonal()))
    xi, iti, resi = orc.pcg(Aop, b, np.zeros(b.size), orc.identity_operator(b.size))
    xc, itc, resc = orc.cg(Aop, b, np.zeros(b.size))
    assert iti == itc and np.array_equal(resi, resc)       # M = I: pcg == cg entry for entry
    u = spla.spsolve(sp.csc_matrix(A), b)
    assert np.linalg.norm(x - u) < 1e-5 * np.linalg.norm(u)


# ------------------------------------------------------------------ golden fixtures
@pytest.mark.parametrize("name", ["micro", "toy"])
def test_oracle_reproduces_golden(orc, fem, name, micro, toy):
    P = {"micro": micro, "toy": toy}[name]
    G = np.load(f"{GOLDEN}/{name}.npz")
    n = P.sub.n_Γ
    assert int(G["n_gamma"]) == n
    assert np.array_equal(G["node_gamma_cnt"], P.sub.node_Γ_cnt)
    for d in range(P.sub.ndom):
        assert np.array_equal(G[f"gather_idx_{d}"], P.sub.gather_idx[d])
        assert np.allclose(G[f"Sd_{d}"], P.Sd[d], rtol=1e-9, atol=1e-12)   # SuperLU pivot order may vary
    assert np.allclose(G["b_schur"], P.b_schur, rtol=1e-10)
    # solver histories from the STORED blocks are reproduced exactly by the oracle
    Sd = [G[f"Sd_{d}"] for d in range(P.sub.ndom)]
    Pi = [G[f"PiSd_{d}"] for d in range(P.sub.ndom)]
    gi = [G[f"gather_idx_{d}"] for d in range(P.sub.ndom)]
    S = orc.apply_local_schurs_operator(Sd, gi, n)
    M = orc.neumann_neumann_operator(Pi, gi, G["node_gamma_cnt"])
    b, x0 = G["b_schur"], np.zeros(n)
    for tag, run in (("cg", lambda: orc.cg(S, b, x0)), ("pcg", lambda: orc.pcg(S, b, x0, M)),
                     ("defpcg", lambda: orc.defpcg(S, b, x0, G["W"], M))):
        x, it, res = run()
        assert it == int(G[f"{tag}_it"])
        assert np.array_equal(res, G[f"{tag}_res_norm"])
        assert np.array_equal(x, G[f"{tag}_x"])


# ------------------------------------------------------------------ reference on-disk formats
def test_reference_npz_round_trip(pkg, fem, tmp_path):
    """save_mesh/load_mesh, save_partition/load_partition in the reference's layout (Fem/Mesh.jl:49-91, 216-247):
    a problem built from the files equals the one built in memory, including on a mesh whose element order
    and neighbour table are permuted (nothing below relies on the structured numbering)."""
    io = pkg.io
    d = str(tmp_path / "data")
    mesh = fem.get_mesh(24)
    epart, npart = fem.mesh_partition(mesh, 2, 2)
    io.save_mesh(mesh, 576, d)
    io.save_partition(epart, npart, 576, 4, d)
    # on disk: cells nel x 3 and 0-based, points nnode x 2, neighbours 1-based with -1 (what set_subdomains expects)
    c = np.load(io.mesh_paths(d, 576)["cells"])
    assert c.shape == (mesh.cells.shape[1], 3) and c.min() == 0
    nb = np.load(io.mesh_paths(d, 576)["cell_neighbors"])
    assert nb.min() == -1 and nb.max() == mesh.cells.shape[1]
    m2 = io.load_mesh(576, d)
    e2, n2 = io.load_partition(576, 4, d)
    for a, b in ((mesh.cells, m2.cells), (mesh.points, m2.points), (mesh.point_marker, m2.point_marker),
                 (mesh.cell_neighbors, m2.cell_neighbors), (epart, e2), (npart, n2)):
        assert np.array_equal(a, b)
    # TriangleMesh's other convention: neighbour indices one too large (EPDD.jl:106-111 corrects it)
    shifted = np.where(mesh.cell_neighbors < 0, 0, mesh.cell_neighbors + 2)
    io._write(io.mesh_paths(d, 576)["cell_neighbors"], shifted.T)
    assert np.array_equal(io.load_mesh(576, d).cell_neighbors, mesh.cell_neighbors)
    P1 = fem.build_schur_problem(24, 2, 2, one, f_m1, u0734)
    P2 = fem.build_schur_problem(24, 2, 2, one, f_m1, u0734, mesh=m2, partition=(e2, n2))
    assert np.array_equal(P1.b_schur, P2.b_schur) and all(np.array_equal(a, b) for a, b in zip(P1.Sd, P2.Sd))
    # a shuffled element order (an unstructured mesh as far as the code can tell)
    perm = np.random.default_rng(0).permutation(mesh.cells.shape[1])
    inv = np.empty_like(perm); inv[perm] = np.arange(perm.size)
    nbp = mesh.cell_neighbors[:, perm]
    m3 = fem.Mesh(mesh.cells[:, perm], mesh.points, mesh.point_marker, np.where(nbp < 0, -1, inv[np.maximum(nbp, 0)]), 24)
    P3 = fem.build_schur_problem(24, 2, 2, one, f_m1, u0734, mesh=m3, partition=(epart[perm], npart))
    assert P3.sub.n_Γ == P1.sub.n_Γ and sorted(P3.sub.n_Γd) == sorted(P1.sub.n_Γd)
    # same Schur system up to the Γ numbering: compare spectra of the assembled operators
    def spectrum(P):
        n = P.sub.n_Γ
        S = np.zeros((n, n))
        for dd in range(P.sub.ndom):
            g = P.sub.gather_idx[dd]
            S[np.ix_(g, g)] += P.Sd[dd]
        return np.linalg.eigvalsh(S)
    assert np.allclose(spectrum(P1), spectrum(P3), rtol=1e-9)
    path = io.save_pcg_iters([15, 16, 15], "DoF576", 4, "0", 3, d)
    assert np.array_equal(np.load(path), [15, 16, 15])


def test_interior_cg_restates_iterative_solvers(orc, micro):
    """`IterativeSolvers.cg(A_IIdd, rhs, reltol=1e-9)` — the reference's interior solve (EPDD.jl:648-650): converges to
    the stated relative residual from x = 0, and the matrix-free operator built on it agrees with the exact operator to
    that tolerance (what Example03:175 prints with its inexact solves)."""
    P = micro
    A = P.A_IIdd[0]
    b = np.random.default_rng(4).standard_normal(A.shape[0])
    for reltol in (1e-9, 1e-4):
        x, it = orc.interior_cg(A, b, reltol)
        assert 0 < it <= A.shape[0]
        assert np.linalg.norm(b - A @ x) <= 1.5 * reltol * np.linalg.norm(b)
    x0, it0 = orc.interior_cg(A, np.zeros(A.shape[0]))
    assert it0 == 0 and not np.any(x0)
    n = P.sub.n_Γ
    Se = orc.apply_local_schurs_operator(P.Sd, P.sub.gather_idx, n)
    Si = orc.apply_local_schurs_matfree_operator(P.A_IIdd, P.A_IΓdd, P.A_ΓΓdd, P.sub.gather_idx, n,
                                                 orc.interior_cg_solvers(P.A_IIdd, 1e-9))
    v = P.b_schur
    assert np.allclose(Si(v), Se(v), rtol=0, atol=1e-7 * np.abs(Se(v)).max())


# ------------------------------------------------------------------ eigCG family, oracle-level pins
def _lap2d(m):
    import scipy.sparse as sp
    k = m - 5                                   # rectangular + anisotropic: simple eigenvalues (a Krylov space sees one
    T = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(m, m))      # vector per eigenspace only)
    U = sp.diags([-1.3, 2.6, -1.3], [-1, 0, 1], shape=(k, k))
    return sp.csc_matrix(sp.kron(sp.identity(k), T) + sp.kron(U, sp.identity(m)) + 0.05 * sp.identity(m * k))


def test_eig_family_known_answers(orc):
    """Properties the algorithms guarantee (no reference fixture exists): the Krylov part of eigcg/eigpcg IS cg/pcg;
    the returned vectors converge to the least-dominant eigenvectors; Def-CG with them needs fewer iterations;
    eigdefpcg keeps r orthogonal to W; Init-CG's first residual is W-orthogonal."""
    A = _lap2d(24)
    n = A.shape[0]
    Ao, Mo = orc.csc_operator(A), orc.jacobi_operator(A.diagonal())
    rng = np.random.default_rng(2)
    b = rng.standard_normal(n)
    x0 = np.zeros(n)
    xc, itc, resc = orc.cg(Ao, b, x0)
    xe, ite, rese, V = orc.eigcg(Ao, b, x0, 6, 20)
    assert abs(ite - itc) <= 1 and np.allclose(rese[:20], resc[:20], rtol=1e-10)
    lam = np.linalg.eigvalsh(A.toarray())[:6]
    rq = np.sort([V[:, j] @ (A @ V[:, j]) / (V[:, j] @ V[:, j]) for j in range(6)])
    assert abs(rq[0] - lam[0]) <= 1e-5 * lam[0] and np.allclose(rq[:3], lam[:3], rtol=3e-2)   # lowest pairs converge first
    assert np.all(rq >= lam * (1 - 1e-10))                  # Ritz values approach from above (interlacing)
    xp, itp, resp = orc.pcg(Ao, b, x0, Mo)
    xq, itq, resq, Vp = orc.eigpcg(Ao, b, x0, Mo, 6, 20)
    assert abs(itq - itp) <= 1 and np.allclose(resq[:20], resp[:20], rtol=1e-10)
    b2 = rng.standard_normal(n)
    it_plain = orc.pcg(Ao, b2, x0, Mo)[1]
    xd, itd, resd, V2 = orc.eigdefpcg(Ao, b2, x0, Mo, Vp, 20)
    assert itd < it_plain and np.linalg.norm(b2 - A @ xd) <= 2e-7 * np.linalg.norm(b2)
    assert orc.defpcg(Ao, b2, x0, Vp, Mo)[1] in (itd - 1, itd, itd + 1)
    xd, itd2, _, V3 = orc.eigdefcg(Ao, b2, x0, V, 20)
    assert itd2 < orc.cg(Ao, b2, x0)[1]
    xi, iti, resi = orc.initcg(Ao, b2, x0, V)
    assert iti <= orc.cg(Ao, b2, x0)[1] and np.linalg.norm(b2 - A @ xi) <= 2e-7 * np.linalg.norm(b2)
    with pytest.raises(orc.BoundsError):
        orc.eigcg(Ao, b, x0, 6, 12)                        # spdim < 2 nvec + 1


def test_host_dense_kernels_of_the_eig_restart(tmp_path):
    """csrc/dense_small.hpp (Jacobi eigen / one-sided Jacobi SVD / the Ritz restart block) compiled for the host."""
    import os
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "dense_small_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "dense_small_check.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout


# ------------------------------------------------------------------ assembly plan: index half on the host
@pytest.mark.parametrize("px,py", [(2, 2), (3, 2)])
def test_assembly_plan_reproduces_prepare_local_schurs(fem, orc, px, py):
    """The plan's (entry -> ordered contributions) structure, executed in numpy, gives bit for bit the blocks and
    right-hand sides of `prepare_local_schurs` for any coefficient — this pins the symbolic half without a GPU."""
    N = 36
    mesh = fem.get_mesh(N)
    dinds = fem.get_dirichlet_inds(mesh.points, mesh.point_marker)
    epart, npart = fem.mesh_partition(mesh, px, py)
    sub = fem.set_subdomains(mesh.cells, mesh.cell_neighbors, epart, npart, dinds.dirichlet_g2l)
    f = lambda x, y: np.sin(3 * x) + y          # noqa: E731
    ue = lambda x, y: 0.5 + x * y               # noqa: E731
    plan = fem.make_assembly_plan(mesh.cells, mesh.points, epart, sub, f, ue)
    for seed in (1, 2):
        a = np.exp(np.random.default_rng(seed).standard_normal(mesh.points.shape[1]))
        want = fem.prepare_local_schurs(mesh.cells, mesh.points, epart, sub, a, f, ue)
        got = plan.blocks(orc.run_assembly_plan(plan, a))
        for k in range(3):
            for A, B in zip(got[k], want[k]):
                assert A.shape == B.shape and np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices)
                assert np.array_equal(A.data, B.data)
        for u, v in zip(got[3], want[3]):
            assert np.array_equal(u, v)
        assert np.array_equal(got[4], want[4])


# ------------------------------------------------------------------ set_subdomains against its loop-by-loop restatement
@pytest.mark.parametrize("case", ["boxes", "strips", "pie5", "pie7"])
def test_set_subdomains_matches_the_reference_loops(fem, orc, case):
    """The vectorised host mirror `fem.set_subdomains` against `oracle.set_subdomains_reference`, a literal transcription
    of EPDD.jl:86-193 (Dicts numbered by first encounter, element/segment/vertex loop order): every map, on box
    partitions and on an unstructured-numbered mesh with pie-slice subdomains meeting at one node (multiplicity 5)."""
    from conftest import unstructured_mesh
    if case == "boxes":
        mesh = fem.get_mesh(19); epart, npart = fem.mesh_partition(mesh, 3, 2)
    elif case == "strips":
        mesh = fem.get_mesh(17); epart, npart = fem.mesh_partition(mesh, 1, 4)
    else:
        mesh, epart, npart = unstructured_mesh(fem, 16 if case == "pie5" else 21, int(case[3:]), seed=3)
    d = fem.get_dirichlet_inds(mesh.points, mesh.point_marker)
    sub = fem.set_subdomains(mesh.cells, mesh.cell_neighbors, epart, npart, d.dirichlet_g2l)
    dirichlet = set(np.flatnonzero(d.dirichlet_g2l >= 0).tolist())
    (ind_Id_g2l, ind_Γd_g2l, ind_Γ_g2l, ind_Γd_Γ2l, owner, elemd, node_Γ, node_Γ_cnt, node_Id, nnode_Id) = \
        orc.set_subdomains_reference(mesh.cells, mesh.cell_neighbors, epart, npart, dirichlet)
    assert np.array_equal(sub.node_Γ, node_Γ) and np.array_equal(sub.node_Γ_cnt, node_Γ_cnt)
    assert np.array_equal(sub.node_owner, owner)
    assert sub.n_Id == nnode_Id
    for dd in range(sub.ndom):
        assert np.array_equal(sub.elemd[dd], elemd[dd]) and np.array_equal(sub.node_Id[dd], node_Id[dd])
        # Γ_d numbering: node_Γd[l] is the node whose Dict value is l; gather_idx is the flattened ind_Γd_Γ2l
        assert {int(g): l for l, g in enumerate(sub.node_Γd[dd])} == ind_Γd_g2l[dd]
        assert {int(g): l for l, g in enumerate(sub.gather_idx[dd])} == ind_Γd_Γ2l[dd]
        assert {int(g): int(sub.ind_I_g2l[g]) for g in sub.node_Id[dd]} == ind_Id_g2l[dd]
    assert {int(g): int(sub.ind_Γ_g2l[g]) for g in sub.node_Γ} == ind_Γ_g2l
    if case.startswith("pie"):
        assert sub.node_Γ_cnt.max() >= 5


def test_assemble_local_schurs_reference_semantics(orc, fem, micro):
    """oracle.assemble_local_schurs (EPDD.jl:667-695: unit vectors through apply_local_schur with the unpreconditioned
    interior cg at reltol 1e-9, upper triangle mirrored) against exact Schur complements: the inexact interior solve shows
    as a relative difference of the order of reltol, and a tighter reltol closes it; prepare_nn is pinv(rtol=sqrt(eps))."""
    P = micro
    Sd9, its = orc.assemble_local_schurs(P.A_IIdd, P.A_IΓdd, P.A_ΓΓdd, reltol=1e-9, return_iterations=True)
    Sd13 = orc.assemble_local_schurs(P.A_IIdd, P.A_IΓdd, P.A_ΓΓdd, reltol=1e-13)
    for d in range(P.sub.ndom):
        exact = P.Sd[d]
        e9 = np.abs(Sd9[d] - exact).max() / np.abs(exact).max()
        e13 = np.abs(Sd13[d] - exact).max() / np.abs(exact).max()
        assert e13 <= 1e-11 and e9 <= 1e-8 and e13 <= e9
        assert np.array_equal(Sd9[d], Sd9[d].T)
        assert its[d] >= P.sub.n_Γd[d]                      # at least one interior iteration per unit vector
    Pi = orc.prepare_neumann_neumann_schur_precond(Sd9)
    for d in range(P.sub.ndom):
        A = Sd9[d]
        assert np.abs(A @ Pi[d] @ A - A).max() <= 1e-7 * np.abs(A).max()
