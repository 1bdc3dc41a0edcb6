"""TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.

ctypes front end of `krylov_oracle.c`, the CPU fp64 restatement of the reference's
Schur-PCG hot path (cg.jl, defcg.jl, EPDD.jl applies). Only tests/, `smoke()` and
`bench.py`'s cpu_baseline leg import this module. PARITY UNPINNED (see the C header).

Function names and argument meaning follow the reference:
  cg(A,b,x;maxit), pcg(A,b,x,M;maxit), defcg(A,b,x,W;maxit), defpcg(A,b,x,W,M;maxit)
    -> (x, it, res_norm[1:it])                     RecyclingKrylovSolvers/cg.jl:14,67; defcg.jl:24,242
All indices are 0-based here.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Callable, Optional, Sequence

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libkrylov_oracle.so")
_lib = None

i64p = C.POINTER(C.c_int64)
f64p = C.POINTER(C.c_double)
INTERIOR_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int64, C.c_int64, f64p, f64p)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "krylov_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libkrylov_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_dot.restype = C.c_double
        L.orc_norm2.restype = C.c_double
        for name in ("orc_csc_op", "orc_diag_op", "orc_schur_assembled_op", "orc_nn_op",
                     "orc_schur_matfree_op", "orc_schur_global_op"):
            getattr(L, name).restype = C.c_void_p
        for name in ("orc_cg", "orc_pcg", "orc_defcg", "orc_defpcg", "orc_interior_cg"):
            getattr(L, name).restype = C.c_int64
        L.orc_op_apply.argtypes = [C.c_void_p, f64p, f64p]
        L.orc_op_free.argtypes = [C.c_void_p]
        L.orc_lu_solve.restype = C.c_int
        L.orc_set_threads.restype = C.c_int
        _lib = L
    return _lib


def set_threads(n: int) -> int:
    return int(lib().orc_set_threads(C.c_int(n)))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _p(a, t):
    return a.ctypes.data_as(t)


def _ptr_array(arrs, t):
    arr = (t * len(arrs))(*[_p(a, t) for a in arrs])
    return arr


class Operator:
    """Anything the solvers may use as `A` (via `A*x`, `mul!`) or `M` (via `M \\ r`)."""

    def __init__(self, handle, n, keep):
        self._h = C.c_void_p(handle)
        self.n = int(n)
        self._keep = keep

    def __call__(self, x):
        x = _f64(x)
        y = np.empty(self.n)
        lib().orc_op_apply(self._h, _p(x, f64p), _p(y, f64p))
        return y

    __mul__ = __call__       # A * x

    def __del__(self):
        try:
            if self._h:
                lib().orc_op_free(self._h)
                self._h = None
        except Exception:
            pass


def csc_operator(A: sp.spmatrix, gather: bool = False) -> Operator:
    """Symmetric SparseMatrixCSC `A` used through `A*x` (stdlib CSC scatter SpMV).
    gather=True uses the row-gather form (same per-row order on symmetric A, threadable)."""
    A = sp.csc_matrix(A)
    A.sort_indices()
    ptr, idx, val = _i64(A.indptr), _i64(A.indices), _f64(A.data)
    n = A.shape[0]
    h = lib().orc_csc_op(C.c_int64(n), _p(ptr, i64p), _p(idx, i64p), _p(val, f64p), C.c_int(int(gather)))
    return Operator(h, n, (ptr, idx, val))


def identity_operator(n: int) -> Operator:
    return Operator(lib().orc_diag_op(C.c_int64(n), None), n, ())


def jacobi_operator(diag: np.ndarray) -> Operator:
    dinv = _f64(1.0 / np.asarray(diag, dtype=np.float64))
    return Operator(lib().orc_diag_op(C.c_int64(dinv.size), _p(dinv, f64p)), dinv.size, (dinv,))


def _dense_blocks(blocks):
    return [np.asfortranarray(np.asarray(b, dtype=np.float64)) for b in blocks]


def apply_local_schurs_operator(Sd: Sequence[np.ndarray], gather_idx: Sequence[np.ndarray], n_Γ: int) -> Operator:
    """`apply_local_schurs(Sd, ind_Γd_Γ2l, node_Γ_cnt, x)` (EPDD.jl:761-785) as an operator."""
    Sd = _dense_blocks(Sd)
    g = [_i64(a) for a in gather_idx]
    nd = _i64([a.size for a in g])
    gp, sp_ = _ptr_array(g, i64p), _ptr_array(Sd, f64p)
    h = lib().orc_schur_assembled_op(C.c_int64(len(Sd)), C.c_int64(n_Γ), _p(nd, i64p), gp, sp_)
    return Operator(h, n_Γ, (Sd, g, nd, gp, sp_))


def neumann_neumann_operator(ΠSd, gather_idx, node_Γ_cnt) -> Operator:
    """`NeumannNeumannSchurPreconditioner(ΠSd, ind_Γd_Γ2l, node_Γ_cnt)` + `\\`
    (EPDD.jl:1111-1137, 1361-1392)."""
    P = _dense_blocks(ΠSd)
    g = [_i64(a) for a in gather_idx]
    nd = _i64([a.size for a in g])
    cnt = _i64(node_Γ_cnt)
    gp, pp = _ptr_array(g, i64p), _ptr_array(P, f64p)
    h = lib().orc_nn_op(C.c_int64(len(P)), C.c_int64(cnt.size), _p(nd, i64p), gp, pp, _p(cnt, i64p))
    return Operator(h, cnt.size, (P, g, nd, cnt, gp, pp))


def _csc_parts(mats):
    ptr, idx, val = [], [], []
    for m in mats:
        m = sp.csc_matrix(m)
        m.sort_indices()
        ptr.append(_i64(m.indptr)); idx.append(_i64(m.indices)); val.append(_f64(m.data))
    return ptr, idx, val


def _wrap_solver(solvers):
    def cb(_user, idom, n, rhs, sol):
        r = np.ctypeslib.as_array(rhs, shape=(n,))
        s = np.ctypeslib.as_array(sol, shape=(n,))
        s[:] = solvers[idom](r.copy())
    return INTERIOR_CB(cb)


def apply_local_schurs_matfree_operator(A_IIdd, A_IΓdd, A_ΓΓdd, gather_idx, n_Γ, solvers) -> Operator:
    """`apply_local_schurs(A_IIdd, A_IΓdd, A_ΓΓdd, ind_Γd_Γ2l, node_Γ_cnt, x)` (EPDD.jl:711-747);
    `solvers[d](rhs)` plays `IterativeSolvers.cg(A_IIdd[d], rhs; ...)` (EPDD.jl:648-650)."""
    g = [_i64(a) for a in gather_idx]
    nd = _i64([a.size for a in g])
    ni = _i64([A.shape[0] for A in A_IIdd])
    igp, igi, igv = _csc_parts(A_IΓdd)
    ggp, ggi, ggv = _csc_parts(A_ΓΓdd)
    cb = _wrap_solver(solvers)
    arrs = (_ptr_array(g, i64p), _ptr_array(igp, i64p), _ptr_array(igi, i64p), _ptr_array(igv, f64p),
            _ptr_array(ggp, i64p), _ptr_array(ggi, i64p), _ptr_array(ggv, f64p))
    h = lib().orc_schur_matfree_op(C.c_int64(len(g)), C.c_int64(n_Γ), _p(nd, i64p), _p(ni, i64p), *arrs, cb, None)
    return Operator(h, n_Γ, (g, nd, ni, igp, igi, igv, ggp, ggi, ggv, cb, arrs, solvers))


def apply_global_schur_operator(A_IId, A_IΓd, A_ΓΓ, solvers) -> Operator:
    """`apply_global_schur(A_IId, A_IΓd, A_ΓΓ, x)` (EPDD.jl:596-625)."""
    n_Γ = A_ΓΓ.shape[0]
    ni = _i64([A.shape[0] for A in A_IId])
    igp, igi, igv = _csc_parts(A_IΓd)
    (ggp,), (ggi,), (ggv,) = _csc_parts([A_ΓΓ])
    cb = _wrap_solver(solvers)
    arrs = (_ptr_array(igp, i64p), _ptr_array(igi, i64p), _ptr_array(igv, f64p))
    h = lib().orc_schur_global_op(C.c_int64(len(A_IId)), C.c_int64(n_Γ), _p(ni, i64p), *arrs,
                                  _p(ggp, i64p), _p(ggi, i64p), _p(ggv, f64p), cb, None)
    return Operator(h, n_Γ, (ni, igp, igi, igv, ggp, ggi, ggv, cb, arrs, solvers))


class SingularException(ArithmeticError):
    """Julia's LinearAlgebra.SingularException from `WtAW \\ mu` (defcg.jl:53, 273)."""


def _solve(fn, A, M, b, x, W, maxit, eps):
    n = A.n
    b = _f64(b)
    x = np.array(x, dtype=np.float64, copy=True)
    res = np.empty(max(n, 1))
    args = [A._h]
    if M is not None:
        args.append(M._h)
    args += [_p(b, f64p), _p(x, f64p)]
    if W is not None:
        W = np.asfortranarray(W, dtype=np.float64)
        args += [_p(W, f64p), C.c_int64(W.shape[1])]
    args += [C.c_int64(maxit), C.c_double(eps), _p(res, f64p)]
    it = int(fn(*args))
    if it == -2 ** 63:
        err = BoundsError(f"res_norm[{n + 1}]: the reference's res_norm has n = {n} entries (cg.jl:23,47; maxit > n only)")
        err.x = x          # the caller's x as the reference had mutated it when it threw
        raise err
    if it < 0:
        raise SingularException(-it)
    return x, it, res[:it].copy()


def cg(A, b, x, maxit=0, eps=1e-7):
    return _solve(lib().orc_cg, A, None, b, x, None, maxit, eps)


def pcg(A, b, x, M, maxit=0, eps=1e-7):
    return _solve(lib().orc_pcg, A, M, b, x, None, maxit, eps)


def defcg(A, b, x, W, maxit=0, eps=1e-7):
    return _solve(lib().orc_defcg, A, None, b, x, W, maxit, eps)


def defpcg(A, b, x, W, M, maxit=0, eps=1e-7):
    return _solve(lib().orc_defpcg, A, M, b, x, W, maxit, eps)


def lu_solve(A, b):
    A = np.asfortranarray(A, dtype=np.float64)
    b = np.array(b, dtype=np.float64, copy=True)
    info = lib().orc_lu_solve(C.c_int64(A.shape[0]), _p(A, f64p), _p(b, f64p))
    if info:
        raise SingularException(info)
    return b


def interior_cg(A: sp.spmatrix, b, reltol: float = 1e-9):
    """`IterativeSolvers.cg(A_IIdd, b, reltol=reltol)` as the reference's matrix-free applies call it
    (EPDD.jl:648-650); returns (x, iterations)."""
    A = sp.csc_matrix(A)
    A.sort_indices()
    ptr, idx, val = _i64(A.indptr), _i64(A.indices), _f64(A.data)
    b = _f64(b)
    x = np.empty(b.size)
    it = lib().orc_interior_cg(C.c_int64(b.size), _p(ptr, i64p), _p(idx, i64p), _p(val, f64p), _p(b, f64p),
                               _p(x, f64p), C.c_double(reltol))
    return x, int(it)


def interior_cg_solvers(A_IIdd, reltol: float = 1e-9):
    """Per-subdomain callables for apply_local_schurs_matfree_operator: the reference's inexact interior solve."""
    return [lambda rhs, A=A: interior_cg(A, rhs, reltol)[0] for A in A_IIdd]


def assemble_local_schurs(A_IIdd, A_IΓdd, A_ΓΓdd, reltol: float = 1e-9, return_iterations: bool = False):
    """`assemble_local_schurs(A_IIdd, A_IΓdd, A_ΓΓdd; preconds, reltol=1e-9)` (EPDD.jl:667-695) in the reference's own
    semantics: column j of S_d is `apply_local_schur` (EPDD.jl:639-654) of the j-th unit vector, its interior solve the
    UNPRECONDITIONED `IterativeSolvers.cg(A_IIdd, ·, reltol=reltol)` from a zero guess (the inverted `isnothing` test
    of :676 means passed preconditioners are ignored), then `Symmetric(·)` (upper triangle mirrored). Returns dense
    column-major blocks (the reference wraps them in `sparse`, which stores every entry)."""
    L = lib()
    out, its = [], []
    for A_II, A_IG, A_GG in zip(A_IIdd, A_IΓdd, A_ΓΓdd):
        ii, ig, gg = (sp.csc_matrix(M) for M in (A_II, A_IG, A_GG))
        for M in (ii, ig, gg):
            M.sort_indices()
        n_i, n_d = ig.shape
        S = np.zeros((n_d, n_d), order="F")
        arrs = []
        for M in (ii, ig, gg):
            arrs += [_i64(M.indptr), _i64(M.indices), _f64(M.data)]
        fn = L.orc_assemble_local_schur
        fn.restype = C.c_int64
        it = fn(C.c_int64(n_i), C.c_int64(n_d), _p(arrs[0], i64p), _p(arrs[1], i64p), _p(arrs[2], f64p),
                _p(arrs[3], i64p), _p(arrs[4], i64p), _p(arrs[5], f64p), _p(arrs[6], i64p), _p(arrs[7], i64p),
                _p(arrs[8], f64p), C.c_double(reltol), _p(S, f64p))
        out.append(S)
        its.append(int(it))
    return (out, its) if return_iterations else out


def prepare_neumann_neumann_schur_precond(Sd):
    """`prepare_neumann_neumann_schur_precond(Sd_local_mat, ind_Γd_Γ2l, node_Γ_cnt)` (EPDD.jl:1201-1220):
    `ΠSd[idom] = LinearAlgebra.pinv(Array(Sd[idom]), rtol = sqrt(eps(Float64)))`. Julia's `pinv(A; rtol)` is the
    SVD `A = U Σ V'` with every σ_i <= rtol·σ_max dropped — numpy's `pinv(rcond)` is the same definition on the same
    LAPACK `gesdd` factorisation (third-party arithmetic: Julia 1.5 stdlib LinearAlgebra, not under /root/reference)."""
    rtol = float(np.sqrt(np.finfo(np.float64).eps))
    return [np.asfortranarray(np.linalg.pinv(np.asarray(S, dtype=np.float64), rcond=rtol)) for S in Sd]


# ---------------------------------------------------------------------------------------------------------------
# eigCG family and Init-CG: numpy restatement, one reference statement per line.
# Dense small-matrix calls follow Julia's LinearAlgebra semantics:
#   eigvecs(Symmetric(T))  -> eigh on the UPPER triangle, ascending
#   rank(Y)                -> #{σ_i > min(m,n)·eps·σ_1}            (stdlib default rtol)
#   svd(Y).U               -> thin U
#   eigen(H)               -> symmetric solver when H is exactly symmetric, else the general one sorted by real part
# The Krylov part (x, it, res_norm) does not depend on any of this for eigcg/eigpcg; the returned V[:, 1:nvec] is
# defined up to the sign (and, for clustered Ritz values, rotation) of eigenvectors, so tests compare subspaces.

class BoundsError(IndexError):
    """Julia's BoundsError where the reference indexes out of range."""


def _julia_rank(Y):
    s = np.linalg.svd(Y, compute_uv=False)
    if s.size == 0 or s[0] == 0.0:
        return 0
    return int(np.sum(s > min(Y.shape) * np.finfo(float).eps * s[0]))


def _eigvecs_sym_upper(T):
    return np.linalg.eigh(np.triu(T) + np.triu(T, 1).T)[1]


def _eigen(H, symmetric):
    if symmetric or np.array_equal(H, H.T):
        return np.linalg.eigh(np.triu(H) + np.triu(H, 1).T)
    vals, Z = np.linalg.eig(H)
    if np.iscomplexobj(vals):
        if np.max(np.abs(vals.imag)) > 0:
            raise TypeError("eigen(H) returned complex eigenvalues (TypeError on the ::Eigen{T,T,...} assertion)")
        vals, Z = vals.real, Z.real
    o = np.argsort(vals, kind="stable")
    return vals[o], Z[:, o]


def _ritz_restart(VtAV, m, nvec, sym_H):
    """eigcg.jl:92-99 / 244-253 / 273-282 on the leading m x m block: returns (vals, Q*Z, nev)."""
    Tm = np.triu(VtAV[:m, :m]) + np.triu(VtAV[:m, :m], 1).T
    Y = np.zeros((m, 2 * nvec))
    Y[:, :nvec] = np.linalg.eigh(Tm)[1][:, :nvec]
    Y[:m - 1, nvec:] = np.linalg.eigh(Tm[:m - 1, :m - 1])[1][:, :nvec]
    nev = _julia_rank(Y)
    Q = np.linalg.svd(Y, full_matrices=False)[0][:, :nev]
    H = Q.T @ (Tm @ Q)
    vals, Z = _eigen(H, sym_H)
    return vals, Q @ Z, nev


def _eig_solver(A, b, x, M, W, nvec, spdim, maxit, eps, kind):
    """Shared body of eigcg (eigcg.jl:27-123), eigpcg (:143-267), eigdefcg (defcg.jl:111-223), eigdefpcg (:337-473)."""
    pre = M is not None
    deflated = W is not None
    n = A.n
    b = _f64(b)
    x = np.array(x, dtype=np.float64, copy=True)
    if deflated:
        W = np.asfortranarray(W, dtype=np.float64)
        nvec = W.shape[1]
    if spdim < 2 * nvec + 1:
        raise BoundsError("ivec = nev + 1 can exceed spdim")
    V = np.zeros((n, spdim), order="F")
    VtAV = np.zeros((spdim, spdim))
    tvec = np.zeros(n)
    res_norm = np.empty(max(n, 1))
    just_restarted = False
    first_restart = True
    hlpr = 0.0
    if maxit == 0:
        maxit = n
    if deflated:
        WtA = np.empty((nvec, n))
        for i in range(nvec):
            WtA[i, :] = A(W[:, i])                                   # defcg.jl:128-131 / 360-363
        WtAW = WtA @ W
        if pre:
            WtW = W.T @ W                                            # defcg.jl:369
        r = b - A(x)
        x += W @ lu_solve(WtAW, W.T @ r)                             # defcg.jl:139-141 / 373-375
    it = 1
    r = b - A(x)
    rTr = r @ r
    z = M(r) if pre else r
    rTz = r @ z if pre else rTr
    if deflated:
        p = z - W @ lu_solve(WtAW, WtA @ z)
        VtAV[:nvec, :nvec] = WtAW
        V[:, :nvec] = W
        ivec = nvec + 1
    else:
        p = z.copy()
        ivec = 1
    res_norm[0] = np.sqrt(rTr)
    V[:, ivec - 1] = z / np.sqrt(rTz) if pre else r / res_norm[0]
    tol = eps * np.sqrt(b @ b)
    while it < maxit and res_norm[it - 1] > tol:
        Ap = A(p)
        d = p @ Ap
        num = rTz if pre else rTr
        alpha = num / d
        beta = 1.0 / num
        x += alpha * p
        r += (-alpha) * Ap
        if deflated and pre:
            r -= W @ lu_solve(WtW, W.T @ r)                          # defcg.jl:411
        rTr = r @ r
        if pre:
            z = M(r)
            if just_restarted and not deflated:
                hlpr = np.sqrt(rTz)                                  # eigcg.jl:212-214
            rTz = r @ z
            beta *= rTz
        else:
            z = r
            beta *= rTr
        if not deflated and ivec == spdim:
            tvec -= beta * Ap                                        # eigcg.jl:71-73 / 217-219
        if deflated:
            p = beta * p + z - W @ lu_solve(WtAW, WtA @ z)
        else:
            p = beta * p + z
        it += 1
        res_norm[it - 1] = np.sqrt(rTr)
        vnew = (lambda: z / np.sqrt(rTz)) if pre else (lambda: r / res_norm[it - 1])

        VtAV[ivec - 1, ivec - 1] += 1.0 / alpha
        if not deflated and just_restarted:
            tvec += Ap
            nev = ivec - 1
            scale = hlpr if pre else res_norm[it - 2]
            VtAV[:nev, ivec - 1] = V[:, :nev].T @ (tvec / scale)     # eigcg.jl:79-84 / 225-230
            just_restarted = False
        if ivec == spdim:
            if deflated:
                if first_restart:
                    VtAV[:nvec, nvec:spdim] = WtA @ V[:, nvec:spdim]  # defcg.jl:186-189 / 422-425
                    first_restart = False
            elif pre:
                AV = np.empty((n, spdim), order="F")
                for j in range(spdim):
                    AV[:, j] = A(V[:, j])                            # eigcg.jl:233-240
                VtAV[:, :] = V.T @ AV
            vals, QZ, nev = _ritz_restart(VtAV, spdim, nvec, sym_H=pre)
            V[:, :nev] = V @ QZ
            ivec = nev + 1
            V[:, ivec - 1] = vnew()
            VtAV[:, :] = 0.0
            VtAV[np.arange(nev), np.arange(nev)] = vals[:nev]
            VtAV[ivec - 1, ivec - 1] = beta / alpha
            if not deflated:
                tvec = -beta * Ap
            just_restarted = True
        else:
            if deflated:
                just_restarted = False                               # defcg.jl:444 (eigdefpcg only; unused in eigdefcg)
            ivec += 1
            V[:, ivec - 1] = vnew()
            VtAV[ivec - 2, ivec - 1] = -np.sqrt(beta) / alpha
            VtAV[ivec - 1, ivec - 1] = beta / alpha
    if pre and not just_restarted:                                   # eigcg.jl:269-287 / defcg.jl:451-470 (pcg variants only)
        if ivec > nvec:
            ivec -= 1
            if deflated and first_restart:
                VtAV[:nvec, nvec:ivec] = WtA @ V[:, nvec:ivec]
            if ivec - 1 < nvec:
                # eigvecs(Tm[1:ivec-1, 1:ivec-1])[:, 1:nvec] (eigcg.jl:275 / defcg.jl:461) is out of range
                raise BoundsError(f"attempt to access {ivec - 1} x {ivec - 1} eigenvector matrix at columns 1:{nvec}")
            vals, QZ, nev = _ritz_restart(VtAV, ivec, nvec, sym_H=True)
            V[:, :nev] = V[:, :ivec] @ QZ
    return x, it, res_norm[:it].copy(), V[:, :nvec].copy()


def eigcg(A, b, x, nvec, spdim, maxit=0, eps=1e-7):
    return _eig_solver(A, b, x, None, None, nvec, spdim, maxit, eps, "eigcg")


def eigpcg(A, b, x, M, nvec, spdim, maxit=0, eps=1e-7):
    return _eig_solver(A, b, x, M, None, nvec, spdim, maxit, eps, "eigpcg")


def eigdefcg(A, b, x, W, spdim, maxit=0, eps=1e-7):
    return _eig_solver(A, b, x, None, W, 0, spdim, maxit, eps, "eigdefcg")


def eigdefpcg(A, b, x, M, W, spdim, maxit=0, eps=1e-7):
    return _eig_solver(A, b, x, M, W, 0, spdim, maxit, eps, "eigdefpcg")


def _init_guess(A, b, x, W):
    """initcg.jl:42-49 / 119-125: x += W (WtAW \\ W'(b - A x)) with WtA = W'A (rows A*W[:,i] on a symmetric A)."""
    W = np.asfortranarray(W, dtype=np.float64)
    x = np.array(x, dtype=np.float64, copy=True)
    WtA = np.empty((W.shape[1], A.n))
    for i in range(W.shape[1]):
        WtA[i, :] = A(W[:, i])
    r = _f64(b) - A(x)
    return x + W @ lu_solve(WtA @ W, W.T @ r)


def initcg(A, b, x, W, maxit=0, eps=1e-7):
    return cg(A, b, _init_guess(A, b, x, W), maxit, eps)


def initpcg(A, b, x, M, W, maxit=0, eps=1e-7):
    return pcg(A, b, _init_guess(A, b, x, W), M, maxit, eps)


# ---------------------------------------------------------------------------------------------------------------
# checker for the device assembly kernel — executes an `fem.AssemblyPlan` in numpy with the
# reference's arithmetic (EPDD.jl:260-269 Δa, :294 ΔK_ij, :322-349 right-hand sides), entry by entry, in order.
def run_assembly_plan(plan, a_nodal):
    a = np.asarray(a_nodal, dtype=np.float64)[plan.cells]
    Δa = np.zeros(plan.cells.shape[1])
    for j in range(3):
        Δa = Δa + a[j]
    Δa = Δa / 3.0
    e, c = plan.ccode // 12, plan.ccode % 12
    stiff = c < 9
    cs = np.where(stiff, c, 0)
    K = Δa[e] * plan.G[cs, e] / 4 / plan.area[e]
    entry = np.repeat(np.arange(plan.n_entries), np.diff(plan.cptr))
    term = np.where(entry < plan.n_matrix_entries, K,
                    np.where(stiff, -(K * plan.ue[cs // 3, e]), plan.be[np.where(stiff, 0, c - 9), e]))
    out = np.zeros(plan.n_entries)
    cnt = np.diff(plan.cptr)
    for p in range(int(cnt.max()) if cnt.size else 0):      # strictly left to right, as `sparse(I,J,V)` / `b[k] +=` do
        m = np.flatnonzero(cnt > p)
        t = term[plan.cptr[m] + p]
        out[m] = t if p == 0 else out[m] + t
    return out


# ------------------------------------------------------------------ set_subdomains, loop by loop (EPDD.jl:86-193)
def set_subdomains_reference(cells, cell_neighbors, epart, npart, dirichlet_nodes):
    """Literal restatement of the reference's `set_subdomains` (Fem/EllipticPdeDomainDecomposition.jl:86-193): the same
    element loop, segment loop and `for node in iel_cell` order, Dicts numbered by first encounter (`.count + 1`),
    the O(n_Γ) `node in node_Γ` membership test, interiors by ascending node id in their `npart` owner. Inputs 0-based
    (cells (3, nel), cell_neighbors with -1 on the boundary, epart, npart; `dirichlet_nodes` a set of 0-based nodes);
    the loops run 1-based internally like the Julia code and the result is returned 0-based:
        ind_Id_g2l, ind_Γd_g2l, ind_Γ_g2l, ind_Γd_Γ2l (lists of dicts / dict), node_owner, elemd, node_Γ, node_Γ_cnt,
        node_Id, nnode_Id
    Pure Python: small meshes only (test infrastructure)."""
    cells = np.asarray(cells) + 1
    nb = np.where(np.asarray(cell_neighbors) < 0, -1, np.asarray(cell_neighbors) + 1)
    epart = np.asarray(epart) + 1
    npart = np.asarray(npart) + 1
    dirichlet = {int(v) + 1 for v in dirichlet_nodes}
    nel = cells.shape[1]
    nnode = int(cells.max())
    ndom = int(epart.max())
    ind_Id_g2l = [dict() for _ in range(ndom)]
    ind_Γd_g2l = [dict() for _ in range(ndom)]
    ind_Γ_g2l = {}
    ind_Γd_Γ2l = [dict() for _ in range(ndom)]
    node_owner = [0] * (nnode + 1)
    elemd = [[] for _ in range(ndom)]
    node_Γ, node_Γ_cnt = [], []
    node_Id = [[] for _ in range(ndom)]
    bnd_tag, iel_max = int(nb.min()), int(nb.max())          # :106-111 (TriangleMesh's off-by-one convention)
    if iel_max > nel:
        nb = nb - 1
        bnd_tag = -1
    for iel in range(1, nel + 1):                              # :114
        iel_cell = [int(v) for v in cells[:, iel - 1]]
        idom = int(epart[iel - 1])
        elemd[idom - 1].append(iel)
        for j in range(3):                                     # :123
            jel = int(nb[j, iel - 1])
            if jel != bnd_tag:
                jdom = int(epart[jel - 1])
                if jdom != idom:
                    jel_cell = [int(v) for v in cells[:, jel - 1]]
                    for node in iel_cell:                      # :135
                        if (node in jel_cell) and (node not in dirichlet):
                            if node not in ind_Γd_g2l[idom - 1]:
                                ind_Γd_g2l[idom - 1][node] = len(ind_Γd_g2l[idom - 1]) + 1
                            if node not in node_Γ:             # :152 (linear search in the reference)
                                node_Γ.append(node)
                                node_Γ_cnt.append(0)
                                ind_Γ_g2l[node] = len(ind_Γ_g2l) + 1
                                node_owner[node] = -1
    for inode in range(1, nnode + 1):                          # :166-173
        if (inode not in dirichlet) and node_owner[inode] != -1:
            idom = int(npart[inode - 1])
            node_Id[idom - 1].append(inode)
            node_owner[inode] = idom
    nnode_Id = [len(v) for v in node_Id]
    for idom in range(ndom):                                   # :176-181
        for i, node in enumerate(node_Id[idom]):
            ind_Id_g2l[idom][node] = i + 1
    for idom in range(ndom):                                   # :184-190
        for gnode, l_in_Γd in ind_Γd_g2l[idom].items():
            l_in_Γ = ind_Γ_g2l[gnode]
            node_Γ_cnt[l_in_Γ - 1] += 1
            ind_Γd_Γ2l[idom][l_in_Γ] = l_in_Γd
    z = lambda d: {k - 1: v - 1 for k, v in d.items()}         # noqa: E731  back to 0-based keys and values
    owner0 = np.array([(-2 if (i in dirichlet) else (-1 if node_owner[i] == -1 else node_owner[i] - 1)) for i in range(1, nnode + 1)])
    return ([z(d) for d in ind_Id_g2l], [z(d) for d in ind_Γd_g2l], z(ind_Γ_g2l), [z(d) for d in ind_Γd_Γ2l], owner0,
            [np.array(e, dtype=np.int64) - 1 for e in elemd], np.array(node_Γ, dtype=np.int64) - 1,
            np.array(node_Γ_cnt, dtype=np.int64), [np.array(v, dtype=np.int64) - 1 for v in node_Id], nnode_Id)
