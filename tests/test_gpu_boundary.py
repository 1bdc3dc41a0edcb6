"""GPU suite (-m gpu): the drop-in boundary as the Julia shim drives it.

  * every `*_create` with 1-based index arrays (`index_base = 1`: gather lists, CSC `colptr`/`rowval`, `cells`) gives
    BITWISE the operator built from 0-based arrays — the shim passes `index_base = 1` everywhere (julia/MI355Schur.jl);
  * tests/c/abi_drive.c: a plain-C program (gcc, no Python, no torch) that links libmi355schur.so and drives it with
    Int64 1-based arrays, host pointers and a C interior-solve callback — the closest executable stand-in for `ccall`;
  * state carried between solves never leaks (a non-finite solve followed by a normal one);
  * `maxit > n` ends where the reference throws BoundsError on `res_norm[n + 1]` (cg.jl:23,47).
"""
import os
import subprocess
import threading

import numpy as np
import pytest

from conftest import ROOT, f_m1, lognormal_coeff, u0734

pytestmark = pytest.mark.gpu


def _one_based(lists):
    return [np.asarray(a, dtype=np.int64) + 1 for a in lists]


def test_every_create_with_one_based_indices(pkg, ctx, orc, fem, ragged):
    api, P = pkg.api, ragged
    n = P.sub.n_Γ
    gi0, gi1, cnt = P.sub.gather_idx, _one_based(P.sub.gather_idx), P.sub.node_Γ_cnt
    rng = np.random.default_rng(21)
    v = rng.standard_normal(n)

    def same(op0, op1, what):
        y0, y1 = op0 * v, op1 * v
        assert np.array_equal(y0, y1), f"{what}: 1-based and 0-based operators differ"
        return y0

    # mi_schur_assembled_create, mi_nn_create
    S0, S1 = api.LocalSchurs(ctx, P.Sd, gi0, cnt), api.LocalSchurs(ctx, P.Sd, gi1, cnt, index_base=1)
    M0 = api.NeumannNeumannSchurPreconditioner(ctx, P.ΠSd, gi0, cnt)
    M1 = api.NeumannNeumannSchurPreconditioner(ctx, P.ΠSd, gi1, cnt, index_base=1)
    same(S0, S1, "mi_schur_assembled_create")
    same(M0, M1, "mi_nn_create")
    r0, r1 = api.pcg(S0, P.b_schur, np.zeros(n), M0), api.pcg(S1, P.b_schur, np.zeros(n), M1)
    assert r0[1] == r1[1] and np.array_equal(r0[2], r1[2]) and np.array_equal(r0[0], r1[0])
    # a sharded slice built from 1-based lists of ALL subdomains (what a Julia rank would pass)
    for lo, hi in ((0, 3), (3, 6)):
        same(api.LocalSchurs(ctx, P.Sd, gi0, cnt, dom_slice=(lo, hi)),
             api.LocalSchurs(ctx, P.Sd, gi1, cnt, index_base=1, dom_slice=(lo, hi)), "assembled slice")

    # mi_schur_matfree_create (host callback) and mi_schur_matfree_device_create
    args = (P.A_IIdd, P.A_IΓdd, P.A_ΓΓdd)
    ym = same(api.MatrixFreeLocalSchurs(ctx, *args, gi0, cnt, P.solvers),
              api.MatrixFreeLocalSchurs(ctx, *args, gi1, cnt, P.solvers, index_base=1), "mi_schur_matfree_create")
    want = orc.apply_local_schurs_matfree_operator(*args, gi0, n, P.solvers) * v
    assert np.array_equal(ym, want)                            # and both are the oracle's bits
    D0 = api.MatrixFreeLocalSchurs(ctx, *args, gi0, cnt, None, reltol=1e-10)
    D1 = api.MatrixFreeLocalSchurs(ctx, *args, gi1, cnt, None, reltol=1e-10, index_base=1)
    same(D0, D1, "mi_schur_matfree_device_create")
    bI = np.concatenate(P.b_Id)
    assert np.array_equal(D0.schur_rhs(bI, P.b_Γ), D1.schur_rhs(bI, P.b_Γ))
    assert np.array_equal(D0.interior_solutions(v, bI), D1.interior_solutions(v, bI))

    # one subdomain in its own numbering (apply_local_schur)
    d = 1
    xd = rng.standard_normal(P.A_ΓΓdd[d].shape[0])
    L0 = api.LocalSchur(ctx, P.A_IIdd[d], P.A_IΓdd[d], P.A_ΓΓdd[d], P.solvers[d])
    L1 = api.LocalSchur(ctx, P.A_IIdd[d], P.A_IΓdd[d], P.A_ΓΓdd[d], P.solvers[d], index_base=1)
    assert np.array_equal(L0 * xd, L1 * xd)

    # mi_schur_global_create / mi_schur_global_device_create
    coeff = lognormal_coeff(fem, P.mesh.points, 7)
    A_IIg, A_IΓg, A_ΓΓ, b_Id, b_Γ = fem.prepare_global_schur(P.mesh.cells, P.mesh.points, P.epart, P.sub, coeff, f_m1, u0734)
    yg = same(api.GlobalSchur(ctx, A_IIg, A_IΓg, A_ΓΓ, P.solvers),
              api.GlobalSchur(ctx, A_IIg, A_IΓg, A_ΓΓ, P.solvers, index_base=1), "mi_schur_global_create")
    assert np.array_equal(yg, orc.apply_global_schur_operator(A_IIg, A_IΓg, A_ΓΓ, P.solvers) * v)
    same(api.GlobalSchur(ctx, A_IIg, A_IΓg, A_ΓΓ, None, reltol=1e-10),
         api.GlobalSchur(ctx, A_IIg, A_IΓg, A_ΓΓ, None, reltol=1e-10, index_base=1), "mi_schur_global_device_create")

    # mi_assembly_plan_create with 1-based `cells`
    plan = fem.make_assembly_plan(P.mesh.cells, P.mesh.points, P.epart, P.sub, f_m1, u0734)
    a = lognormal_coeff(fem, P.mesh.points, 3)
    v0, v1 = api.AssemblyPlan(ctx, plan).run(a), api.AssemblyPlan(ctx, plan, index_base=1).run(a)
    assert np.array_equal(v0, v1) and np.array_equal(v0, orc.run_assembly_plan(plan, a))

    # out-of-range indices in either convention are refused, not wrapped around
    bad = [g.copy() for g in gi1]
    bad[0][0] = 0                                              # 0 is not a 1-based index
    with pytest.raises(pkg._lib.MiError):
        api.LocalSchurs(ctx, P.Sd, bad, cnt, index_base=1)
    bad[0][0] = n + 1
    with pytest.raises(pkg._lib.MiError):
        api.LocalSchurs(ctx, P.Sd, bad, cnt, index_base=1)


def test_plain_c_driver(pkg, orc, micro, tmp_path):
    """tests/c/abi_drive.c compiled with gcc against include/mi355schur.h and libmi355schur.so; the problem goes over in
    a flat binary file (Int64 1-based index arrays, column-major blocks), the results come back the same way and are
    compared with the oracle."""
    P = micro
    n, ndom = P.sub.n_Γ, P.sub.ndom
    exe = str(tmp_path / "abi_drive")
    lib_dir = os.path.dirname(pkg._lib.LIB_PATH)
    subprocess.check_call(["gcc", "-O1", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_drive.c"), "-o", exe, "-L", lib_dir, "-lmi355schur",
                           "-lm", f"-Wl,-rpath,{lib_dir}"])
    # ---- problem file: header, then per subdomain the arrays a Julia caller holds (1-based Int64)
    import scipy.sparse as sp
    inp, out = str(tmp_path / "problem.bin"), str(tmp_path / "result.bin")
    with open(inp, "wb") as f:
        np.array([ndom, n], dtype=np.int64).tofile(f)
        np.asarray(P.sub.node_Γ_cnt, dtype=np.int64).tofile(f)
        P.b_schur.astype(np.float64).tofile(f)
        for d in range(ndom):
            nd, ni = P.A_ΓΓdd[d].shape[0], P.A_IIdd[d].shape[0]
            np.array([nd, ni], dtype=np.int64).tofile(f)
            (np.asarray(P.sub.gather_idx[d], dtype=np.int64) + 1).tofile(f)
            np.asfortranarray(P.Sd[d]).ravel(order="F").tofile(f)
            np.asfortranarray(P.ΠSd[d]).ravel(order="F").tofile(f)
            for A in (P.A_IΓdd[d], P.A_ΓΓdd[d]):
                A = sp.csc_matrix(A); A.sort_indices()
                np.array([A.nnz], dtype=np.int64).tofile(f)
                (A.indptr.astype(np.int64) + 1).tofile(f)
                (A.indices.astype(np.int64) + 1).tofile(f)
                A.data.astype(np.float64).tofile(f)
            # the C callback does the interior solve with a dense inverse handed over by the test (column-major)
            np.asfortranarray(np.linalg.inv(P.A_IIdd[d].toarray())).ravel(order="F").tofile(f)
    r = subprocess.run([exe, inp, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = np.fromfile(out, dtype=np.float64)
    it_pcg, it_cg = int(raw[0]), int(raw[1])
    pos = 2
    x_pcg = raw[pos:pos + n]; pos += n
    res_pcg = raw[pos:pos + it_pcg]; pos += it_pcg
    y_S = raw[pos:pos + n]; pos += n
    y_M = raw[pos:pos + n]; pos += n
    y_mf = raw[pos:pos + n]; pos += n
    x_cg = raw[pos:pos + n]; pos += n
    assert pos == raw.size
    So = orc.apply_local_schurs_operator(P.Sd, P.sub.gather_idx, n)
    Mo = orc.neumann_neumann_operator(P.ΠSd, P.sub.gather_idx, P.sub.node_Γ_cnt)
    xo, ito, reso = orc.pcg(So, P.b_schur, np.zeros(n), Mo)
    assert it_pcg == ito and np.allclose(res_pcg, reso, rtol=1e-8, atol=1e-12 * reso[0])
    assert np.linalg.norm(x_pcg - xo) <= 1e-6 * np.linalg.norm(xo)
    assert np.allclose(y_S, So * P.b_schur, rtol=0, atol=1e-13 * np.abs(y_S).max())
    assert np.allclose(y_M, Mo * P.b_schur, rtol=0, atol=1e-13 * np.abs(y_M).max())
    # matrix-free apply through the C callback (dense inverse): the Example03:175 identity against the assembled apply
    assert np.allclose(y_mf, y_S, rtol=0, atol=1e-9 * np.abs(y_S).max())
    xc, itc, _ = orc.cg(So, P.b_schur, np.zeros(n))
    assert abs(it_cg - itc) <= max(1, itc // 50) and np.linalg.norm(x_cg - xc) <= 1e-4 * np.linalg.norm(xc)
    assert "abi_drive ok" in r.stdout


def test_non_finite_solve_does_not_poison_the_next(pkg, ctx, orc, toy):
    """A solve with NaN in b ends at once (res_norm[1] = NaN fails `> tol`); the next solve on the SAME operators and
    workspace must be unaffected (the reference carries no state between solves)."""
    api, P = pkg.api, toy
    n = P.sub.n_Γ
    S = api.LocalSchurs(ctx, P.Sd, P.sub.gather_idx, P.sub.node_Γ_cnt)
    M = api.NeumannNeumannSchurPreconditioner(ctx, P.ΠSd, P.sub.gather_idx, P.sub.node_Γ_cnt)
    good = api.pcg(S, P.b_schur, np.zeros(n), M)
    bad_b = P.b_schur.copy(); bad_b[n // 2] = np.nan
    xb, itb, resb = api.pcg(S, bad_b, np.zeros(n), M)
    assert itb == 1 and np.isnan(resb[0])
    inf_b = P.b_schur.copy(); inf_b[3] = np.inf
    api.pcg(S, inf_b, np.zeros(n), M)
    again = api.pcg(S, P.b_schur, np.zeros(n), M)
    assert again[1] == good[1] and np.array_equal(again[2], good[2]) and np.array_equal(again[0], good[0])
    for solver in (lambda b: api.cg(S, b, np.zeros(n), maxit=12), lambda b: api.pcg(S, b, np.ones(n), M)):
        ref = solver(P.b_schur); solver(bad_b); got = solver(P.b_schur)
        assert got[1] == ref[1] and np.array_equal(got[2], ref[2]) and np.array_equal(got[0], ref[0])


def test_maxit_beyond_n_stops_where_the_reference_throws(pkg, ctx, orc):
    """maxit > n with an unreachable tolerance: the reference throws BoundsError at `res_norm[n + 1] = ...` after n loop
    iterations (cg.jl:23,47). Same stop, same x, on every loop form."""
    import scipy.sparse as sp
    api = pkg.api
    n = 9
    rng = np.random.default_rng(4)
    Q = rng.standard_normal((n, n))
    A = sp.csr_matrix(Q @ Q.T + n * np.eye(n))
    b = rng.standard_normal(n)
    Ao, Io = orc.csc_operator(A), orc.identity_operator(n)
    Ad, Id = api.SparseMatrixCSC(ctx, A), api.IdentityPreconditioner(ctx, n)
    for dev, ref in ((lambda: api.cg(Ad, b, np.zeros(n), maxit=3 * n, eps=1e-300),
                      lambda: orc.cg(Ao, b, np.zeros(n), maxit=3 * n, eps=1e-300)),
                     (lambda: api.pcg(Ad, b, np.zeros(n), Id, maxit=3 * n, eps=1e-300),
                      lambda: orc.pcg(Ao, b, np.zeros(n), Io, maxit=3 * n, eps=1e-300))):
        with pytest.raises(orc.BoundsError) as eo:
            ref()
        with pytest.raises(api.BoundsError) as eg:
            dev()
        assert eg.value.it == n + 1
        assert np.allclose(eg.value.x, eo.value.x, rtol=1e-9, atol=1e-12)
    # maxit == n is legal and fills res_norm exactly
    x, it, res = api.cg(Ad, b, np.zeros(n), maxit=n, eps=1e-300)
    xo, ito, reso = orc.cg(Ao, b, np.zeros(n), maxit=n, eps=1e-300)
    assert it == ito == n and res.size == n


def test_loopback_allreduce_entry_point_and_global_schur_host_pointers(pkg, orc, fem, ragged):
    """mi_ctx_allreduce_sum on in-process ranks really sums (it used to return its input on a loopback context); and a
    host-pointer apply of the global Schur operator (read-modify-write passes on y) equals the device-pointer apply."""
    import torch
    api = pkg.api
    world = 3
    group = api.LoopbackGroup(world)
    out = [None] * world

    def rank(r):
        c = api.Context(0)
        c.loopback_init(group, r)
        out[r] = c.allreduce_sum(np.arange(5, dtype=np.float64) * (r + 1))
    ts = [threading.Thread(target=rank, args=(r,)) for r in range(world)]
    [t.start() for t in ts]; [t.join() for t in ts]
    for r in range(world):
        assert np.array_equal(out[r], np.arange(5) * 6.0)
    P = ragged
    coeff = lognormal_coeff(fem, P.mesh.points, 7)
    A_IIg, A_IΓg, A_ΓΓ, _, _ = fem.prepare_global_schur(P.mesh.cells, P.mesh.points, P.epart, P.sub, coeff, f_m1, u0734)
    c = api.Context(0)
    Sg = api.GlobalSchur(c, A_IIg, A_IΓg, A_ΓΓ, P.solvers)
    v = np.random.default_rng(2).standard_normal(P.sub.n_Γ)
    yd = Sg.apply(torch.from_numpy(v).cuda()); c.synchronize()
    assert np.array_equal(Sg * v, yd.cpu().numpy())
