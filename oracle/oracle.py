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
