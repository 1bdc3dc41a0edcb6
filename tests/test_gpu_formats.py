"""GPU suite, SURVEY.md §8 f4 end to end: a mesh and a partition written in the reference's on-disk layout
(Fem/Mesh.jl:49-55 save_mesh, :216-219 save_partition — 0-based `cells`, transposed arrays, NPY payload under an
`.npz` name) are loaded through io.py, turned into operators and solved on the device, against the oracle.

The files hold what Triangle + METIS would produce as far as any code can tell (conftest.unstructured_mesh): jittered
nodes, permuted node and element numbers, pie-slice subdomains that meet at one node. That node has multiplicity 5
or 6, so the contribution-slot table is that wide: the folded 2-launch loop does not apply (slot width <= 4) and the generic
4-launch loop runs — the first non-box partition to reach the kernels."""
import numpy as np
import pytest

from conftest import f_m1, u0734, unstructured_mesh

pytestmark = pytest.mark.gpu


def test_unstructured_mesh_files_to_device_solve(pkg, ctx, orc, fem, tmp_path):
    from test_gpu_parity import assert_history
    api, io = pkg.api, pkg.io
    d = str(tmp_path / "data")
    nsec, tent = 6, 1600
    mesh0, epart0, npart0 = unstructured_mesh(fem, 40, nsec, seed=5)
    io.save_mesh(mesh0, tent, d)
    io.save_partition(epart0, npart0, tent, nsec, d)
    mesh = io.load_mesh(tent, d)
    epart, npart = io.load_partition(tent, nsec, d)
    assert np.array_equal(mesh.cells, mesh0.cells) and np.array_equal(mesh.cell_neighbors, mesh0.cell_neighbors)
    coeff = lambda x, y: 1.0 + 0.5 * np.sin(5 * x) * np.cos(3 * y)      # noqa: E731
    P = fem.build_schur_problem(40, 0, 0, coeff, f_m1, u0734, mesh=mesh, partition=(epart, npart))
    sub = P.sub
    n = sub.n_Γ
    assert sub.ndom == nsec and sub.node_Γ_cnt.max() >= 5               # the hub: more sharers than any box partition has
    S = api.LocalSchurs(ctx, P.Sd, sub.gather_idx, sub.node_Γ_cnt)
    M = api.NeumannNeumannSchurPreconditioner(ctx, P.ΠSd, sub.gather_idx, sub.node_Γ_cnt)
    So = orc.apply_local_schurs_operator(P.Sd, sub.gather_idx, n)
    Mo = orc.neumann_neumann_operator(P.ΠSd, sub.gather_idx, sub.node_Γ_cnt)
    v = np.random.default_rng(8).standard_normal(n)
    ys, ym = S * v, M.ldiv(v)
    assert np.allclose(ys, So * v, rtol=0, atol=1e-13 * np.abs(ys).max())
    assert np.allclose(ym, Mo * v, rtol=0, atol=1e-13 * np.abs(ym).max())
    # matrix-free operators on the same files: bit-exact with the oracle (host callback), Example03:175 identity
    Sm = api.MatrixFreeLocalSchurs(ctx, P.A_IIdd, P.A_IΓdd, P.A_ΓΓdd, sub.gather_idx, sub.node_Γ_cnt, P.solvers)
    Smo = orc.apply_local_schurs_matfree_operator(P.A_IIdd, P.A_IΓdd, P.A_ΓΓdd, sub.gather_idx, n, P.solvers)
    assert np.array_equal(Sm * v, Smo * v)
    assert np.allclose(Sm * v, ys, rtol=0, atol=1e-10 * np.abs(ys).max())
    # solvers
    x0 = np.zeros(n)
    got = api.pcg(S, P.b_schur, x0, M)
    assert_history(got, orc.pcg(So, P.b_schur, x0, Mo), apply=So, b=P.b_schur)
    assert_history(api.cg(S, P.b_schur, x0), orc.cg(So, P.b_schur, x0), apply=So, b=P.b_schur)
    Sdense = np.column_stack([So * e for e in np.eye(n)])
    W = np.asfortranarray(np.linalg.eigh((Sdense + Sdense.T) / 2)[1][:, :nsec + 10])
    assert_history(api.defpcg(S, P.b_schur, x0, W, M), orc.defpcg(So, P.b_schur, x0, W, Mo), apply=So, b=P.b_schur)
    # Example03:204: Schur solution + back-substitution = direct solve of the full system on the same mesh
    u_Γ = got[0]
    u_I = Sm.interior_solutions(u_Γ, np.concatenate(P.b_Id))
    offs = np.cumsum([0] + sub.n_Id)
    u = fem.merge_subdomain_solutions(u_Γ, [u_I[offs[k]:offs[k + 1]] for k in range(nsec)], sub, P.dinds, u0734, mesh.points)
    dinds = fem.get_dirichlet_inds(mesh.points, mesh.point_marker)
    A, b = fem.do_isotropic_elliptic_assembly(mesh.cells, mesh.points, dinds, mesh.point_marker, coeff, f_m1, u0734)
    import scipy.sparse.linalg as spla
    u_full = fem.append_bc(dinds, spla.spsolve(A.tocsc(), b), mesh.points, u0734)
    assert np.abs(u - u_full).max() <= 1e-6 * np.abs(u_full).max()
    # iteration counts go back out in the reference's file layout (Example07:281-285)
    path = io.save_pcg_iters([got[1]], f"DoF{tent}", nsec, "0", 1, d)
    assert np.array_equal(np.load(path), [got[1]])
