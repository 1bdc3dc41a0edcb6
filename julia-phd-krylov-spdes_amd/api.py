"""Host-side mirror of the reference's operator / preconditioner / solver interface for the
Schur-PCG hot path, on top of the C ABI (include/mi355schur.h). Same names, argument order and
return values as the Julia functions; citations are relative to /root/reference
("EPDD.jl" = Fem/EllipticPdeDomainDecomposition.jl).

    cg(A, b, x; maxit=0)            -> (x, it, res_norm[1:it])     RecyclingKrylovSolvers/cg.jl:14
    pcg(A, b, x, M; maxit=0)                                        cg.jl:67
    defcg(A, b, x, W; maxit=0)                                      defcg.jl:24
    defpcg(A, b, x, W, M; maxit=0)                                  defcg.jl:242
    apply_local_schurs(S, x)  /  S * x                              EPDD.jl:711-785
    apply_global_schur(S, x)                                        EPDD.jl:596-625
    NeumannNeumannSchurPreconditioner(ΠSd, ind_Γd_Γ2l, node_Γ_cnt)  EPDD.jl:1111-1137
    apply_neumann_neumann_schur(Πnn, r)  /  Πnn.ldiv(r)             EPDD.jl:1361-1403

Vectors may be numpy arrays (host pointers; copied through the boundary like Julia arrays) or
torch CUDA tensors (device pointers, zero copy). Indices are 0-based (`index_base=0`); a Julia
caller passes `index_base=1` through the shim in julia/MI355Schur.jl.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, List, Optional, Sequence

import numpy as np
import scipy.sparse as sp

from . import _lib
from ._lib import (BoundsError, MiError, SingularException, check, f64p, f64pp, i64, i64p, i64pp, vp)  # noqa: F401

EPS = 1e-7  # RecyclingKrylovSolvers.jl:21 `const eps = 1e-7`


def _is_torch(x) -> bool:
    t = type(x)
    return t is not np.ndarray and t.__module__.startswith("torch")


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _ptrs(arrs, elem_ptr_type):
    """C array of pointers; None entries become NULL (other ranks' subdomains)."""
    out = (elem_ptr_type * len(arrs))()
    for k, a in enumerate(arrs):
        out[k] = a.ctypes.data_as(elem_ptr_type) if a is not None else None
    return out


class Context:
    """One GPU + one HIP stream (+ an optional RCCL communicator). `mi_ctx_t`."""

    def __init__(self, device: int = 0):
        L = _lib.load()
        h = vp()
        check(L.mi_ctx_create(C.c_int(device), C.byref(h)))
        self._h, self._L, self.device = h, L, device
        self.rank, self.n_ranks = 0, 1

    # -- pointer mode follows the argument type of each call
    def _mode_for(self, *vecs) -> int:
        dev = [v for v in vecs if v is not None and _is_torch(v)]
        if dev and len(dev) != len([v for v in vecs if v is not None]):
            raise TypeError("mix of torch tensors and numpy arrays in one call")
        mode = _lib.MI_PTR_DEVICE if dev else _lib.MI_PTR_HOST
        check(self._L.mi_ctx_set_pointer_mode(self._h, mode))
        return mode

    def _res_buffer(self, cap: int):
        """Landing zone of a solve's residual history, kept per context (one allocation and one address lookup instead of
        one per solve: `ndarray.ctypes` alone costs several microseconds, a fifth of what a 15-iteration solve's launch does)."""
        buf = getattr(self, "_res", None)
        if buf is None or buf[0].size < cap:
            a = np.empty(max(cap, 64))
            buf = self._res = (a, C.cast(a.ctypes.data, f64p))
        return buf

    @staticmethod
    def _ptr(v, n: Optional[int] = None, writable: bool = False):
        """(keepalive, void*) of a contiguous fp64 vector."""
        if v is None:
            return None, None
        if _is_torch(v):
            import torch
            if v.dtype != torch.float64 or not v.is_cuda or not v.is_contiguous():
                raise TypeError("device vectors must be contiguous float64 CUDA tensors")
            if n is not None and v.numel() != n:
                raise ValueError(f"expected {n} entries, got {v.numel()}")
            return v, vp(v.data_ptr())
        a = v if (isinstance(v, np.ndarray) and v.dtype == np.float64 and v.flags.c_contiguous
                  and (v.flags.writeable or not writable)) else _f64(v)
        if n is not None and a.size != n:
            raise ValueError(f"expected {n} entries, got {a.size}")
        return a, vp(a.ctypes.data)

    def loopback_init(self, group: "LoopbackGroup", rank: int) -> None:
        """Make this context rank `rank` of an in-process group (one host thread per rank): the single-GPU stand-in for
        `comm_init`, used to test the sharded operators (`mi_ctx_loopback_init`)."""
        check(self._L.mi_ctx_loopback_init(self._h, group._h, C.c_int(rank)))
        self.rank, self.n_ranks, self._group = rank, group.n, group

    def set_chunk(self, iterations_per_graph: int) -> None:
        check(self._L.mi_ctx_set_chunk(self._h, C.c_int(iterations_per_graph)))

    def use_torch_stream(self) -> None:
        """Launch on torch's current stream (so torch.cuda.Event sees our kernels, and tensors produced by torch kernels
        just before a call are ordered with it). Not the legacy default stream: the solvers capture graphs."""
        import torch
        check(self._L.mi_ctx_set_stream(self._h, vp(torch.cuda.current_stream(self.device).cuda_stream)))

    def use_own_stream(self) -> None:
        check(self._L.mi_ctx_set_stream(self._h, None))

    def _record(self, *tensors) -> None:
        """Asynchronous entry points (`mi_dense_set_blocks`, `mi_schur_setup_run`, `mi_nn_pinv`, `mi_assembly_run` in
        device-pointer mode) return while the context's stream still reads / writes the caller's torch tensors. Torch's
        current stream is made to wait for the context's stream at this point (an event, no host synchronisation), so a
        tensor that dies right after the call is not handed out again — on torch's stream — before the work is done.
        (Not `Tensor.record_stream`: the allocator would record an event on the context's stream when the tensor is freed,
        possibly after the context — and its stream — are gone.) Contract: INTEGRATION.md "device pointers"."""
        self._order(tensors, after=True)

    def _order(self, tensors, after: bool) -> None:
        """The Python operators that hand torch tensors back (`A * x`, `schur_rhs`, `interior_solutions`) behave like torch
        ops: the context's stream waits for torch's current stream before the call, torch's current stream waits for the
        context's after it (two events, no host synchronisation), and the allocator is told (`_record`). The bare C entry
        points keep the contract of INTEGRATION.md: ordering is the caller's."""
        ts = [t for t in tensors if t is not None and _is_torch(t) and t.is_cuda]
        if not ts:
            return
        import torch
        sp = vp()
        check(self._L.mi_ctx_get_stream(self._h, C.byref(sp)))
        cur = torch.cuda.current_stream(ts[0].device)
        if not sp.value or cur.cuda_stream == sp.value:
            return
        ext = torch.cuda.ExternalStream(sp.value, device=ts[0].device)
        if after:
            cur.wait_stream(ext)
        else:
            ext.wait_stream(cur)

    def synchronize(self) -> None:
        check(self._L.mi_ctx_synchronize(self._h))

    # -- RCCL
    def unique_id(self) -> bytes:
        buf = C.create_string_buffer(_lib.MI_COMM_ID_BYTES)
        check(self._L.mi_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, uid: bytes, rank: int, n_ranks: int) -> None:
        buf = C.create_string_buffer(uid, _lib.MI_COMM_ID_BYTES)
        check(self._L.mi_ctx_comm_init(self._h, buf, C.c_int(rank), C.c_int(n_ranks)))
        self.rank, self.n_ranks = rank, n_ranks

    # -- the one-shot peer exchange (include/mi355schur.h; `@distributed (+)`, EllipticPdePllDomainDecomposition.jl:10-14)
    def peer_init(self, rank: int, n_ranks: int, arena_bytes: int = 0) -> None:
        check(self._L.mi_ctx_peer_init(self._h, C.c_int(rank), C.c_int(n_ranks), i64(arena_bytes)))
        self.rank, self.n_ranks = rank, n_ranks

    def peer_export(self):
        """(handle bytes for other processes, arena base address for contexts of this process)"""
        buf = C.create_string_buffer(_lib.MI_PEER_HANDLE_BYTES)
        base = vp()
        check(self._L.mi_ctx_peer_export(self._h, buf, C.byref(base)))
        return buf.raw, int(base.value or 0)

    def peer_import(self, rank: int, handle: bytes = None, same_process_base: int = 0) -> None:
        buf = C.create_string_buffer(handle, _lib.MI_PEER_HANDLE_BYTES) if handle is not None else None
        check(self._L.mi_ctx_peer_import(self._h, C.c_int(rank), buf, vp(same_process_base or None)))

    def peer_ready(self) -> None:
        check(self._L.mi_ctx_peer_ready(self._h))

    def set_exchange(self, mode) -> None:
        """`mi_ctx_set_exchange`: 0 / False — RCCL for everything; 1 / True — peer exchange, one-wave wait kernels between the
        launches; 2 — peer exchange, the folded launches wait for the flags themselves (only with one GPU per rank: a
        waiting launch keeps its compute units)."""
        check(self._L.mi_ctx_set_exchange(self._h, C.c_int(int(mode))))

    def query(self, what: str) -> int:
        """`mi_ctx_query`: "no_graph", "peer_exchange", "graph_replays", "exchanges", "spectral_pinv"."""
        out = i64(0)
        check(self._L.mi_ctx_query(self._h, C.c_int({"no_graph": 0, "peer_exchange": 1, "graph_replays": 2, "exchanges": 3, "spectral_pinv": 4, "experimental": 5}[what]), C.byref(out)))
        return int(out.value)

    def peer_connect(self, rank: int, n_ranks: int, all_gather) -> None:
        """Whole hand-shake over any transport: `all_gather(obj) -> list of every rank's obj` (e.g. a wrapper of
        torch.distributed.all_gather_object)."""
        self.peer_init(rank, n_ranks)
        handle, _ = self.peer_export()
        handles = all_gather(handle)
        for q, h in enumerate(handles):
            if q != rank:
                self.peer_import(q, h)
        self.peer_ready()

    def allreduce_sum(self, v):
        self._mode_for(v)
        keep, p = self._ptr(v, writable=True)
        n = keep.numel() if _is_torch(keep) else keep.size
        check(self._L.mi_ctx_allreduce_sum(self._h, p, i64(n)))
        return keep

    # -- BLAS-1 (RecyclingKrylovSolvers.jl:3)
    def dot(self, x, y) -> float:
        self._mode_for(x, y)
        kx, px = self._ptr(x)
        n = kx.numel() if _is_torch(kx) else kx.size
        ky, py = self._ptr(y, n)
        out = C.c_double()
        check(self._L.mi_dot(self._h, i64(n), px, py, C.byref(out)))
        return out.value

    def norm2(self, x) -> float:
        self._mode_for(x)
        kx, px = self._ptr(x)
        n = kx.numel() if _is_torch(kx) else kx.size
        out = C.c_double()
        check(self._L.mi_norm2(self._h, i64(n), px, C.byref(out)))
        return out.value

    def axpy(self, a: float, x, y):
        """axpy!(a, x, y): y += a*x, returns y (numpy input: a new array unless y is writable fp64)."""
        self._mode_for(x, y)
        ky, py = self._ptr(y, writable=True)
        n = ky.numel() if _is_torch(ky) else ky.size
        kx, px = self._ptr(x, n)
        check(self._L.mi_axpy(self._h, i64(n), C.c_double(a), px, py))
        return ky

    def axpby(self, a: float, x, b: float, y):
        """axpby!(a, x, b, y): y = a*x + b*y."""
        self._mode_for(x, y)
        ky, py = self._ptr(y, writable=True)
        n = ky.numel() if _is_torch(ky) else ky.size
        kx, px = self._ptr(x, n)
        check(self._L.mi_axpby(self._h, i64(n), C.c_double(a), px, C.c_double(b), py))
        return ky

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._L.mi_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LoopbackGroup:
    """`mi_loopback_group_create`: n in-process "ranks" (test facility, see include/mi355schur.h)."""

    def __init__(self, n: int):
        L = _lib.load()
        h = vp()
        check(L.mi_loopback_group_create(C.c_int(n), C.byref(h)))
        self._h, self._L, self.n = h, L, n

    def set_mode(self, mode: int) -> None:
        """0: device-side peer exchange when GPU_MAX_HW_QUEUES >= n (graphs); 1: host rendezvous (eager launches)."""
        check(self._L.mi_loopback_group_set_mode(self._h, C.c_int(mode)))

    def __del__(self):
        try:
            if self._h:
                self._L.mi_loopback_group_destroy(self._h)
                self._h = None
        except Exception:
            pass


class Event:
    def __init__(self, ctx: Context):
        self.ctx = ctx
        self._h = vp()
        check(ctx._L.mi_event_create(C.byref(self._h)))

    def record(self) -> "Event":
        check(self.ctx._L.mi_event_record(self.ctx._h, self._h))
        return self

    def elapsed_ms(self, stop: "Event") -> float:
        out = C.c_double()
        check(self.ctx._L.mi_event_elapsed_ms(self._h, stop._h, C.byref(out)))
        return out.value

    def __del__(self):
        try:
            if self._h:
                self.ctx._L.mi_event_destroy(self._h)
                self._h = None
        except Exception:
            pass


class Operator:
    """`mi_op_t`: anything the solvers use through `A*x` / `mul!` or `M \\ r`."""

    def __init__(self, ctx: Context, handle, keep=()):
        self.ctx, self._h, self._keep = ctx, handle, keep
        n = i64()
        check(ctx._L.mi_op_size(handle, C.byref(n)))
        self.n = self.N = int(n.value)   # `.N` as LinearMaps.FunctionMap exposes it (Example07:273)

    def apply(self, x, out=None):
        self.ctx._mode_for(x, out)
        kx, px = self.ctx._ptr(x, self.n)
        if out is None:
            if _is_torch(kx):
                import torch
                out = torch.empty_like(kx)
            else:
                out = np.empty(self.n)
        ko, po = self.ctx._ptr(out, self.n, writable=True)
        self.ctx._order((kx, ko), after=False)
        check(self.ctx._L.mi_op_apply(self._h, px, po))
        self.ctx._order((kx, ko), after=True)
        return ko

    __call__ = apply
    __mul__ = apply          # A * x
    ldiv = apply             # M \ r  (`\` has no Python spelling)

    def bytes(self):
        a, d = i64(), i64()
        check(self.ctx._L.mi_op_bytes(self._h, C.byref(a), C.byref(d)))
        return int(a.value), int(d.value)

    def apply_dominant(self, x, reps: int = 1) -> None:
        self.ctx._mode_for(x)
        kx, px = self.ctx._ptr(x, self.n)
        check(self.ctx._L.mi_op_apply_dominant(self._h, px, C.c_int(reps)))

    def time_dominant(self, x, reps: int = 200) -> float:
        """Average microseconds per launch of the dominant kernel (HIP events, graph replay)."""
        self.ctx._mode_for(x)
        kx, px = self.ctx._ptr(x, self.n)
        out = C.c_double()
        check(self.ctx._L.mi_op_time_dominant(self._h, px, C.c_int(reps), C.byref(out)))
        return out.value

    def close(self) -> None:
        if getattr(self, "_h", None) and getattr(self.ctx, "_h", None):
            self.ctx._L.mi_op_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------ operator constructors
def SparseMatrixCSC(ctx: Context, A, index_base: int = 0) -> Operator:
    """A symmetric `SparseMatrixCSC{Float64,Int}` as the solvers' `A` (cg.jl:14-15).
    `A` is a scipy sparse matrix or a (colptr, rowval, nzval, n) tuple."""
    if isinstance(A, tuple):
        ptr, idx, val, n = A
        ptr, idx, val = _i64(ptr), _i64(idx), _f64(val)
    else:
        A = sp.csr_matrix(A)
        A.sort_indices()
        n = A.shape[0]
        if A.shape[0] != A.shape[1]:
            raise ValueError("square matrix expected")
        ptr, idx, val = _i64(A.indptr), _i64(A.indices), _f64(A.data)
    h = vp()
    check(ctx._L.mi_csr_create(ctx._h, i64(n), i64(n), ptr.ctypes.data_as(i64p), idx.ctypes.data_as(i64p),
                               val.ctypes.data_as(f64p), C.c_int(index_base), C.byref(h)))
    return Operator(ctx, h)


def IdentityPreconditioner(ctx: Context, n: int) -> Operator:
    h = vp()
    check(ctx._L.mi_diag_create(ctx._h, i64(n), None, C.byref(h)))
    return Operator(ctx, h)


def JacobiPreconditioner(ctx: Context, diag) -> Operator:
    """z = r ./ diag(A) — stands in for Example01's AMG preconditioner (out of scope)."""
    dinv = _f64(1.0 / np.asarray(diag, dtype=np.float64))
    h = vp()
    check(ctx._L.mi_diag_create(ctx._h, i64(dinv.size), dinv.ctypes.data_as(f64p), C.byref(h)))
    return Operator(ctx, h)


def _dom_slice(ctx: Context, ndom: int, dom_slice):
    if dom_slice is None:
        return shard_domains(ndom, ctx.rank, ctx.n_ranks)
    return int(dom_slice[0]), int(dom_slice[1])


def shard_domains(ndom: int, rank: int, n_ranks: int):
    """Contiguous block of subdomains owned by `rank` (8 subdomains: 8/4/2/1 per GPU at 1/2/4/8 GPUs)."""
    lo = ndom * rank // n_ranks
    hi = ndom * (rank + 1) // n_ranks
    return lo, hi


def _blocks(blocks, lo, hi):
    out = []
    for d, b in enumerate(blocks):
        out.append(np.asfortranarray(np.asarray(b, dtype=np.float64)) if (lo <= d < hi and b is not None) else None)
    return out


class LocalSchurs(Operator):
    """Assembled Schur operator: `x -> apply_local_schurs(Sd, ind_Γd_Γ2l, node_Γ_cnt, x)` (EPDD.jl:761-785),
    i.e. the closure Example03:131-135 wraps in a LinearMap."""

    def __init__(self, ctx: Context, Sd: Sequence, ind_Γd_Γ2l: Sequence, node_Γ_cnt, index_base: int = 0,
                 dom_slice=None):
        ndom = len(Sd)
        n_Γ = len(node_Γ_cnt)
        lo, hi = _dom_slice(ctx, ndom, dom_slice)
        g = [_i64(a) for a in ind_Γd_Γ2l]
        nd = _i64([a.size for a in g])
        S = _blocks(Sd, lo, hi)
        for d in range(lo, hi):
            if S[d].shape != (nd[d], nd[d]):
                raise ValueError(f"Sd[{d}] has shape {S[d].shape}, expected {(nd[d], nd[d])}")
        h = vp()
        check(ctx._L.mi_schur_assembled_create(ctx._h, i64(ndom), i64(n_Γ), nd.ctypes.data_as(i64p), _ptrs(g, i64p),
                                               _ptrs(S, f64p), C.c_int(index_base), i64(lo), i64(hi), C.byref(h)))
        super().__init__(ctx, h)
        self.dom_slice = (lo, hi)


class NeumannNeumannSchurPreconditioner(Operator):
    """`NeumannNeumannSchurPreconditioner(ΠSd, ind_Γd_Γ2l, node_Γ_cnt)` (EPDD.jl:1111-1137); used by the
    solvers through `Πnn \\ r` (EPDD.jl:1389-1392) = `apply_neumann_neumann_schur` (EPDD.jl:1361-1386)."""

    def __init__(self, ctx: Context, ΠSd: Sequence, ind_Γd_Γ2l: Sequence, node_Γ_cnt, index_base: int = 0,
                 dom_slice=None):
        ndom = len(ΠSd)
        cnt = _i64(node_Γ_cnt)
        lo, hi = _dom_slice(ctx, ndom, dom_slice)
        g = [_i64(a) for a in ind_Γd_Γ2l]
        nd = _i64([a.size for a in g])
        P = _blocks(ΠSd, lo, hi)
        h = vp()
        check(ctx._L.mi_nn_create(ctx._h, i64(ndom), i64(cnt.size), nd.ctypes.data_as(i64p), _ptrs(g, i64p),
                                  _ptrs(P, f64p), cnt.ctypes.data_as(i64p), C.c_int(index_base), i64(lo), i64(hi),
                                  C.byref(h)))
        super().__init__(ctx, h)
        self.dom_slice = (lo, hi)


def _wrap_interior(solvers: Sequence[Callable]):
    """`solvers[d](rhs) -> A_II[d]^{-1} rhs` on the host, as the C callback (EPDD.jl:648-650)."""
    def cb(_user, idom, n, rhs, sol):
        try:
            r = np.ctypeslib.as_array(rhs, shape=(n,))
            s = np.ctypeslib.as_array(sol, shape=(n,))
            s[:] = solvers[idom](r.copy())
            return 0
        except Exception:  # never let an exception cross the C frame
            return 1
    return _lib.INTERIOR_SOLVE_FN(cb)


def _csc_parts(mats, lo, hi, base: int = 0):
    """CSC arrays of the blocks lo..hi-1 as int64 in the `base` convention (1: what Julia's `colptr`/`rowval` hold)."""
    ptr, idx, val = [], [], []
    for d, m in enumerate(mats):
        if lo <= d < hi:
            m = sp.csc_matrix(m)
            m.sort_indices()
            ptr.append(_i64(m.indptr) + base); idx.append(_i64(m.indices) + base); val.append(_f64(m.data))
        else:
            ptr.append(None); idx.append(None); val.append(None)
    return ptr, idx, val


class MatrixFreeLocalSchurs(Operator):
    """`x -> apply_local_schurs(A_IIdd, A_IΓdd, A_ΓΓdd, ind_Γd_Γ2l, node_Γ_cnt, x; preconds, reltol)` (EPDD.jl:711-747),
    the closure of Example03:143-150. Sparse products on the device; `A_IIdd^{-1}` either through
    `interior_solvers[d](rhs)` on the host, or — `interior_solvers=None` — by the device CG that restates the
    reference's own `IterativeSolvers.cg(A_IIdd, rhs, reltol=reltol)` (EPDD.jl:648-650)."""

    def __init__(self, ctx: Context, A_IIdd, A_IΓdd, A_ΓΓdd, ind_Γd_Γ2l, node_Γ_cnt, interior_solvers=None,
                 reltol: float = 1e-9, dom_slice=None, index_base: int = 0):
        """`index_base=1`: `ind_Γd_Γ2l` holds 1-based Γ indices and the CSC arrays cross the ABI 1-based, exactly as the
        Julia shim passes `colptr`/`rowval` (the matrices themselves are scipy objects either way)."""
        ndom = len(A_IΓdd)
        n_Γ = len(node_Γ_cnt)
        lo, hi = _dom_slice(ctx, ndom, dom_slice)
        g = [_i64(a) for a in ind_Γd_Γ2l]
        nd = _i64([a.size for a in g])
        ni = _i64([A.shape[0] for A in A_IIdd])
        igp, igi, igv = _csc_parts(A_IΓdd, lo, hi, index_base)
        ggp, ggi, ggv = _csc_parts(A_ΓΓdd, lo, hi, index_base)
        h = vp()
        if interior_solvers is None:
            iip, iii, iiv = _csc_parts(A_IIdd, lo, hi, index_base)
            check(ctx._L.mi_schur_matfree_device_create(
                ctx._h, i64(ndom), i64(n_Γ), nd.ctypes.data_as(i64p), ni.ctypes.data_as(i64p), _ptrs(g, i64p),
                _ptrs(iip, i64p), _ptrs(iii, i64p), _ptrs(iiv, f64p), _ptrs(igp, i64p), _ptrs(igi, i64p), _ptrs(igv, f64p),
                _ptrs(ggp, i64p), _ptrs(ggi, i64p), _ptrs(ggv, f64p), C.c_double(reltol), C.c_int(index_base), i64(lo), i64(hi),
                C.byref(h)))
            super().__init__(ctx, h)
            return
        cb = _wrap_interior(interior_solvers)
        check(ctx._L.mi_schur_matfree_create(
            ctx._h, i64(ndom), i64(n_Γ), nd.ctypes.data_as(i64p), ni.ctypes.data_as(i64p), _ptrs(g, i64p),
            _ptrs(igp, i64p), _ptrs(igi, i64p), _ptrs(igv, f64p), _ptrs(ggp, i64p), _ptrs(ggi, i64p), _ptrs(ggv, f64p),
            cb, None, C.c_int(index_base), i64(lo), i64(hi), C.byref(h)))
        super().__init__(ctx, h, keep=(cb, interior_solvers))

    def set_values(self, ii_val=None, ig_val=None, gg_val=None) -> None:
        """New block values on the patterns given at construction (Example07:162-171 without re-creating the operator).
        Each argument: the concatenation over this operator's subdomains of the blocks' CSC `nzval` — the layout of
        `AssemblyPlan.run` / `fem.AssemblyPlan.layout` — as numpy arrays or torch CUDA tensors; None = unchanged."""
        self.ctx._mode_for(ii_val, ig_val, gg_val)
        k1, p1 = self.ctx._ptr(ii_val)
        k2, p2 = self.ctx._ptr(ig_val)
        k3, p3 = self.ctx._ptr(gg_val)
        check(self.ctx._L.mi_schur_matfree_set_values(self._h, p1, p2, p3))

    def schur_rhs(self, b_I, b_Γ):
        """`get_schur_rhs(b_Id, A_IIdd, A_IΓdd, b_Γ, ind_Γd_Γ2l; preconds)` (EPDD.jl:835-864) with this operator's
        interior solve; `b_I` is the concatenation of the b_Id of this operator's subdomains."""
        self.ctx._mode_for(b_I, b_Γ)
        k1, p1 = self.ctx._ptr(b_I)
        k2, p2 = self.ctx._ptr(b_Γ, self.n)
        if _is_torch(b_Γ):
            import torch
            out = torch.empty_like(b_Γ)
            po = vp(out.data_ptr())
        else:
            out = np.empty(self.n)
            po = vp(out.ctypes.data)
        self.ctx._order((k1, k2, out), after=False)
        check(self.ctx._L.mi_schur_matfree_rhs(self._h, p1, p2, po))
        self.ctx._order((k1, k2, out), after=True)
        return out


def _interior_solutions(self, u_Γ, b_I):
    """`get_subdomain_solutions(u_Γ, A_IId, A_IΓd, b_Id)` (EPDD.jl:1014-1025) with this operator's interior solve:
    the concatenation of u_Id = A_IIdd \\ (b_Id - A_IΓdd u_Γd) over this operator's subdomains."""
    self.ctx._mode_for(u_Γ, b_I)
    k1, p1 = self.ctx._ptr(u_Γ, self.n)
    k2, p2 = self.ctx._ptr(b_I)
    if _is_torch(b_I):
        import torch
        out = torch.empty_like(b_I)
        po = vp(out.data_ptr())
    else:
        out = np.empty(np.asarray(b_I).size)
        po = vp(out.ctypes.data)
    self.ctx._order((k1, k2, out), after=False)
    check(self.ctx._L.mi_schur_matfree_interior_solutions(self._h, p1, p2, po))
    self.ctx._order((k1, k2, out), after=True)
    return out


MatrixFreeLocalSchurs.interior_solutions = _interior_solutions


def _interior_precond(self, kind) -> None:
    """The `precond(s)` keyword of `apply_local_schur(s)` / `apply_global_schur` (EPDD.jl:648-650, 609-619) for the device
    interior CG: None / "none": plain CG (the default); "diagonal": `Pl = Diagonal(A_IIdd)`."""
    k = {None: 0, "none": 0, 0: 0, "diagonal": 1, "jacobi": 1, 1: 1}.get(kind)
    if k is None:
        raise ValueError("interior preconditioner: None or 'diagonal'")
    check(self.ctx._L.mi_schur_interior_precond(self._h, C.c_int(k)))


def _interior_iterations(self) -> int:
    """Diagnostic: iterations the device interior CG has run so far (slowest subdomain, rounded up to whole replays)."""
    out = i64(0)
    check(self.ctx._L.mi_schur_interior_iterations(self._h, C.byref(out)))
    return int(out.value)


MatrixFreeLocalSchurs.interior_precond = _interior_precond
MatrixFreeLocalSchurs.interior_iterations = _interior_iterations


class LocalSchur(MatrixFreeLocalSchurs):
    """`xd -> apply_local_schur(A_IIdd, A_IΓdd, A_ΓΓdd, xd; precond, reltol)` (EPDD.jl:639-654): ONE subdomain, vectors in
    its own Γ_d numbering — S_d xd = A_ΓΓdd xd - A_IΓdd' A_IIdd^{-1} (A_IΓdd xd). What `assemble_local_schurs` applies to the
    unit vectors (EPDD.jl:667-695) and `prepare_neumann_neumann_schur_precond` wraps (:1152-1189)."""

    def __init__(self, ctx: Context, A_IIdd, A_IΓdd, A_ΓΓdd, interior_solver=None, reltol: float = 1e-9,
                 index_base: int = 0):
        n = A_ΓΓdd.shape[0]
        super().__init__(ctx, [A_IIdd], [A_IΓdd], [A_ΓΓdd], [np.arange(n, dtype=np.int64) + index_base],
                         np.ones(n, dtype=np.int64), None if interior_solver is None else [interior_solver], reltol,
                         index_base=index_base)


class AssemblyPlan:
    """Device executor of a `fem.AssemblyPlan` (`mi_plan_t`): `run(a)` is the numeric half of
    `prepare_local_schurs(cells, points, epart, ..., a, f, uexact)` (EPDD.jl:389-546) for a new coefficient vector."""

    def __init__(self, ctx: Context, plan, index_base: int = 0):
        self.ctx, self.plan = ctx, plan
        h = vp()
        cells = _i64(plan.cells) + index_base      # index_base=1: node numbers as Julia's `cells` holds them
        arrs = [_f64(plan.G), _f64(plan.area), _f64(plan.ue), _f64(plan.be)]
        cptr, ccode = _i64(plan.cptr), _i64(plan.ccode)
        check(ctx._L.mi_assembly_plan_create(
            ctx._h, i64(cells.shape[1]), i64(plan.n_node), cells.ctypes.data_as(i64p), C.c_int(index_base),
            *[a.ctypes.data_as(f64p) for a in arrs], i64(plan.n_entries), i64(plan.n_matrix_entries),
            cptr.ctypes.data_as(i64p), ccode.ctypes.data_as(i64p), C.byref(h)))
        self._h = h

    def run(self, a_nodal):
        """values[n_entries]: numpy in -> numpy out; torch CUDA tensor in -> torch CUDA tensor out (stays on the device)."""
        self.ctx._mode_for(a_nodal)
        ka, pa = self.ctx._ptr(a_nodal, self.plan.n_node)
        if _is_torch(a_nodal):
            import torch
            out = torch.empty(self.plan.n_entries, dtype=torch.float64, device=a_nodal.device)
            po = vp(out.data_ptr())
        else:
            out = np.empty(self.plan.n_entries)
            po = vp(out.ctypes.data)
        check(self.ctx._L.mi_assembly_run(self._h, pa, po))
        self.ctx._record(a_nodal, out)
        return out

    def block_values(self, values, lo: int = 0, hi: Optional[int] = None):
        """(ii_val, ig_val, gg_val, b_I, b_Γ) slices of `values` for subdomains lo..hi-1: contiguous views, ready for
        `MatrixFreeLocalSchurs.set_values` / `.schur_rhs`."""
        L = self.plan.layout
        hi = len(L["II"]) if hi is None else hi

        def span(name):
            return values[L[name][lo][0]:L[name][hi - 1][0] + L[name][hi - 1][1]]
        off, cnt = L["bΓ"]
        return span("II"), span("IΓ"), span("ΓΓ"), span("bI"), values[off:off + cnt]

    def close(self) -> None:
        if getattr(self, "_h", None) and getattr(self.ctx, "_h", None):
            self.ctx._L.mi_assembly_plan_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SchurSetup:
    """`assemble_local_schurs(A_IIdd, A_IΓdd, A_ΓΓdd, ...)` (EPDD.jl:667-695) on the device (`mi_setup_t`): built once from the
    blocks' sparsity, `run` per realization with the blocks' values — numpy arrays, or torch CUDA tensors as
    `AssemblyPlan.run` / `.block_values` leave them (nothing visits the host then)."""

    def __init__(self, ctx: Context, A_IIdd, A_IΓdd, A_ΓΓdd, index_base: int = 0):
        self.ctx = ctx
        ndom = len(A_IΓdd)
        self.n_Γd = [int(A.shape[0]) for A in A_ΓΓdd]
        self.n_Id = [int(A.shape[0]) for A in A_IIdd]
        nd, ni = _i64(self.n_Γd), _i64(self.n_Id)
        iip, iii, iiv = _csc_parts(A_IIdd, 0, ndom, index_base)
        igp, igi, igv = _csc_parts(A_IΓdd, 0, ndom, index_base)
        ggp, ggi, ggv = _csc_parts(A_ΓΓdd, 0, ndom, index_base)
        self._vals = (np.concatenate(iiv) if iiv else np.empty(0), np.concatenate(igv), np.concatenate(ggv))
        h = vp()
        check(ctx._L.mi_schur_setup_create(ctx._h, i64(ndom), nd.ctypes.data_as(i64p), ni.ctypes.data_as(i64p),
                                           _ptrs(iip, i64p), _ptrs(iii, i64p), _ptrs(igp, i64p), _ptrs(igi, i64p),
                                           _ptrs(ggp, i64p), _ptrs(ggi, i64p), C.c_int(index_base), C.byref(h)))
        self._h = h
        self.n_S = int(sum(n * n for n in self.n_Γd))
        self.n_w = int(sum(self.n_Γd))

    def run(self, ii_val=None, ig_val=None, gg_val=None, b_I=None):
        """-> (Sd, w): the concatenated column-major S_d blocks and (with b_I) the concatenated w_d = A_IΓdd' (A_IIdd \\ b_Id).
        Values default to those of the matrices given at construction."""
        ii_val = self._vals[0] if ii_val is None else ii_val
        ig_val = self._vals[1] if ig_val is None else ig_val
        gg_val = self._vals[2] if gg_val is None else gg_val
        self.ctx._mode_for(ii_val, ig_val, gg_val, b_I)
        k1, p1 = self.ctx._ptr(ii_val)
        k2, p2 = self.ctx._ptr(ig_val)
        k3, p3 = self.ctx._ptr(gg_val)
        k4, p4 = self.ctx._ptr(b_I)
        if _is_torch(ig_val):
            import torch
            Sd = torch.empty(self.n_S, dtype=torch.float64, device=ig_val.device)
            w = torch.empty(self.n_w, dtype=torch.float64, device=ig_val.device) if b_I is not None else None
            pS, pw = vp(Sd.data_ptr()), (vp(w.data_ptr()) if w is not None else None)
        else:
            Sd = np.empty(self.n_S)
            w = np.empty(self.n_w) if b_I is not None else None
            pS, pw = vp(Sd.ctypes.data), (vp(w.ctypes.data) if w is not None else None)
        check(self.ctx._L.mi_schur_setup_run(self._h, p1, p2, p3, p4, pS, pw))
        self.ctx._record(ii_val, ig_val, gg_val, b_I, Sd, w)
        return Sd, w

    def keep_levels(self, on: bool = True) -> None:
        """Every following `run` keeps the inverses of all elimination levels (`mi_schur_setup_keep_levels`): exact interior
        solves `interior_solve(f)` and `MatrixFreeLocalSchurs.use_level_solver(self)`."""
        check(self.ctx._L.mi_schur_setup_keep_levels(self._h, C.c_int(1 if on else 0)))

    def interior_solve(self, f):
        """u = A_IIdd \\ f for all subdomains (concatenated interior vectors, the order of b_I): exact, on the device."""
        self.ctx._mode_for(f)
        n = int(sum(self.n_Id))
        kf, pf = self.ctx._ptr(f, n)
        if _is_torch(f):
            import torch
            u = torch.empty_like(f)
            pu = vp(u.data_ptr())
        else:
            u = np.empty(n)
            pu = vp(u.ctypes.data)
        check(self.ctx._L.mi_schur_setup_interior_solve(self._h, pf, pu))
        self.ctx._record(f, u)
        return u

    def blocks(self, Sd):
        """The list of n_Γd x n_Γd column-major blocks of a concatenated buffer (views)."""
        out, off = [], 0
        for n in self.n_Γd:
            b = Sd[off:off + n * n]
            out.append(b.reshape(n, n).T if not _is_torch(Sd) else b.view(n, n).T)
            off += n * n
        return out

    def close(self) -> None:
        if getattr(self, "_h", None) and getattr(self.ctx, "_h", None):
            self.ctx._L.mi_schur_setup_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def nn_pinv(ctx: Context, n_Γd, Sd, rtol: float = 0.0):
    """`prepare_neumann_neumann_schur_precond(Sd, ...)`'s numeric half (EPDD.jl:1211): the concatenated ΠS_d = pinv(S_d,
    rtol = sqrt(eps)) of the concatenated symmetric blocks, on the device (`mi_nn_pinv`)."""
    nd = _i64(n_Γd)
    ctx._mode_for(Sd)
    k, p = ctx._ptr(Sd, int((nd * nd).sum()))
    if _is_torch(Sd):
        import torch
        out = torch.empty_like(Sd)
        po = vp(out.data_ptr())
    else:
        out = np.empty(k.size)
        po = vp(out.ctypes.data)
    check(ctx._L.mi_nn_pinv(ctx._h, i64(nd.size), nd.ctypes.data_as(i64p), p, C.c_double(rtol), po))
    ctx._record(Sd, out)
    return out


def device_dense_setup(ctx: Context):
    """`dense_setup` hook of `fem.build_schur_problem`: S_d, w_d (`assemble_local_schurs` + the condensed rhs of `get_schur_rhs`,
    EPDD.jl:667-695, 853-861) and ΠS_d (`prepare_neumann_neumann_schur_precond`, :1201-1220) from this library's device set-up
    (`mi_schur_setup_run`, `mi_nn_pinv`) — host arrays in and out."""
    def run(A_II, A_IΓ, A_ΓΓ, b_Id, want_pinv):
        setup = SchurSetup(ctx, A_II, A_IΓ, A_ΓΓ)
        Sd, w = setup.run(b_I=np.concatenate(b_Id))
        Sl = [np.asfortranarray(b) for b in setup.blocks(Sd)]
        wl, off = [], 0
        for n in setup.n_Γd:
            wl.append(w[off:off + n].copy())
            off += n
        Pl = [np.asfortranarray(b) for b in setup.blocks(nn_pinv(ctx, setup.n_Γd, Sd))] if want_pinv else None
        setup.close()
        return Sl, wl, Pl
    return run


def _set_blocks(self, blocks):
    """New S_d / ΠS_d (one concatenated column-major buffer, numpy or torch CUDA) on this operator's maps (`mi_dense_set_blocks`)."""
    self.ctx._mode_for(blocks)
    k, p = self.ctx._ptr(blocks)
    check(self.ctx._L.mi_dense_set_blocks(self._h, p))
    self.ctx._record(blocks)


def _use_level_solver(self, setup: "SchurSetup" = None):
    """Interior solves of this matrix-free / global Schur operator through the level inverses `setup` keeps (exact, device;
    `mi_schur_matfree_interior_levels`); None: back to the operator's own callback / interior CG."""
    check(self.ctx._L.mi_schur_matfree_interior_levels(self._h, setup._h if setup is not None else None))
    self._level_setup = setup      # keep the plan alive


LocalSchurs.set_blocks = _set_blocks
NeumannNeumannSchurPreconditioner.set_blocks = _set_blocks
MatrixFreeLocalSchurs.use_level_solver = _use_level_solver


class GlobalSchur(Operator):
    """`x -> apply_global_schur(A_IId, A_IΓd, A_ΓΓ, x; preconds)` (EPDD.jl:596-625), the closure of Example03:101.
    `interior_solvers[d](rhs)` on the host, or — `interior_solvers=None` — the device CG that restates the reference's
    `IterativeSolvers.cg(A_IId[idom], A_IΓd[idom]*x)` (EPDD.jl:609-619; package default reltol = sqrt(eps))."""

    def __init__(self, ctx: Context, A_IId, A_IΓd, A_ΓΓ, interior_solvers=None, reltol: float = float(np.sqrt(np.finfo(float).eps)),
                 index_base: int = 0):
        ndom = len(A_IΓd)
        n_Γ = A_ΓΓ.shape[0]
        ni = _i64([A.shape[0] for A in A_IId])
        igp, igi, igv = _csc_parts(A_IΓd, 0, ndom, index_base)
        (ggp,), (ggi,), (ggv,) = _csc_parts([A_ΓΓ], 0, 1, index_base)
        h = vp()
        if interior_solvers is None:
            iip, iii, iiv = _csc_parts(A_IId, 0, ndom, index_base)
            check(ctx._L.mi_schur_global_device_create(
                ctx._h, i64(ndom), i64(n_Γ), ni.ctypes.data_as(i64p), _ptrs(iip, i64p), _ptrs(iii, i64p), _ptrs(iiv, f64p),
                _ptrs(igp, i64p), _ptrs(igi, i64p), _ptrs(igv, f64p), ggp.ctypes.data_as(i64p), ggi.ctypes.data_as(i64p),
                ggv.ctypes.data_as(f64p), C.c_double(reltol), C.c_int(index_base), C.byref(h)))
            super().__init__(ctx, h)
            return
        cb = _wrap_interior(interior_solvers)
        check(ctx._L.mi_schur_global_create(
            ctx._h, i64(ndom), i64(n_Γ), ni.ctypes.data_as(i64p), _ptrs(igp, i64p), _ptrs(igi, i64p), _ptrs(igv, f64p),
            ggp.ctypes.data_as(i64p), ggi.ctypes.data_as(i64p), ggv.ctypes.data_as(f64p), cb, None, C.c_int(index_base), C.byref(h)))
        super().__init__(ctx, h, keep=(cb, interior_solvers))


GlobalSchur.interior_solutions = _interior_solutions      # `get_subdomain_solutions` (EPDD.jl:1014-1025), Γ-global columns
GlobalSchur.schur_rhs = MatrixFreeLocalSchurs.schur_rhs   # `get_schur_rhs` (EPDD.jl:798-821)
GlobalSchur.interior_precond = _interior_precond
GlobalSchur.interior_iterations = _interior_iterations


# reference-named free functions
def apply_local_schur(S_d: LocalSchur, xd):
    return S_d.apply(xd)


def apply_local_schurs(S: Operator, x):
    return S.apply(x)


def apply_global_schur(S: GlobalSchur, x):
    return S.apply(x)


def apply_neumann_neumann_schur(Πnn: NeumannNeumannSchurPreconditioner, r):
    return Πnn.apply(r)


# ------------------------------------------------------------------ solvers
def _solve(kind: str, A: Operator, M: Optional[Operator], b, x, W, maxit: int, eps: float, nvec_out: int = 0,
           spdim: int = 0):
    ctx = A.ctx
    n = A.n
    ctx._mode_for(b, x, W)
    kb, pb = ctx._ptr(b, n)
    if _is_torch(x):
        kx, px = ctx._ptr(x, n, writable=True)       # mutated in place like the reference's x
    else:
        kx = np.array(x, dtype=np.float64, copy=True)
        if kx.size != n:
            raise ValueError(f"expected {n} entries, got {kx.size}")
        px = vp(kx.ctypes.data)
    cap = int(min(maxit if maxit else n, n)) + 1
    res, res_p = ctx._res_buffer(cap)
    it = i64()
    L = ctx._L
    tail = (maxit, eps, res_p, cap, C.byref(it))
    if W is not None:
        if _is_torch(W):
            if W.dim() != 2 or W.shape[0] != n or W.stride(0) != 1:
                raise TypeError("W must be an n x nvec column-major tensor (e.g. torch.empty(nvec, n).T)")
            nvec, kW, pW = W.shape[1], W, vp(W.data_ptr())
        else:
            kW = np.asfortranarray(W, dtype=np.float64)
            if kW.ndim != 2 or kW.shape[0] != n:
                raise ValueError("W must be n x nvec")
            nvec, pW = kW.shape[1], vp(kW.ctypes.data)
    V = pV = None
    if kind.startswith("eig"):
        nv = nvec if W is not None else int(nvec_out)
        # V[:, 1:nvec] comes back where x lives: a column-major torch tensor on the device, or a Fortran numpy array
        if _is_torch(x):
            import torch
            V = torch.empty((nv, n), dtype=torch.float64, device=x.device).T
            pV = vp(V.data_ptr())
        else:
            V = np.empty((n, nv), order="F")
            pV = vp(V.ctypes.data)
    if kind == "cg":
        rc = L.mi_cg(A._h, pb, px, *tail)
    elif kind == "pcg":
        rc = L.mi_pcg(A._h, M._h, pb, px, *tail)
    elif kind == "defcg":
        rc = L.mi_defcg(A._h, pb, px, pW, i64(nvec), *tail)
    elif kind == "defpcg":
        rc = L.mi_defpcg(A._h, M._h, pb, px, pW, i64(nvec), *tail)
    elif kind == "initcg":
        rc = L.mi_initcg(A._h, pb, px, pW, i64(nvec), *tail)
    elif kind == "initpcg":
        rc = L.mi_initpcg(A._h, M._h, pb, px, pW, i64(nvec), *tail)
    elif kind == "eigcg":
        rc = L.mi_eigcg(A._h, pb, px, i64(nvec_out), i64(spdim), *tail, pV)
    elif kind == "eigpcg":
        rc = L.mi_eigpcg(A._h, M._h, pb, px, i64(nvec_out), i64(spdim), *tail, pV)
    elif kind == "eigdefcg":
        rc = L.mi_eigdefcg(A._h, pb, px, pW, i64(nvec), i64(spdim), *tail, pV)
    elif kind == "eigdefpcg":
        rc = L.mi_eigdefpcg(A._h, M._h, pb, px, pW, i64(nvec), i64(spdim), *tail, pV)
    else:
        raise ValueError(kind)
    try:
        check(rc)
    except BoundsError as e:   # the reference mutates x in place before `res_norm[it]` throws: hand the state over
        e.x, e.it = kx, int(it.value)
        raise
    k = int(it.value)
    if V is not None:
        return kx, k, res[:k].copy(), V
    return kx, k, res[:k].copy()


def cg(A: Operator, b, x, maxit: int = 0, eps: float = EPS):
    """cg(A, b, x; maxit=0) (cg.jl:14-50)."""
    return _solve("cg", A, None, b, x, None, maxit, eps)


def pcg(A: Operator, b, x, M: Operator, maxit: int = 0, eps: float = EPS):
    """pcg(A, b, x, M; maxit=0) (cg.jl:67-109)."""
    return _solve("pcg", A, M, b, x, None, maxit, eps)


def defcg(A: Operator, b, x, W, maxit: int = 0, eps: float = EPS):
    """defcg(A, b, x, W; maxit=0) (defcg.jl:24-83)."""
    return _solve("defcg", A, None, b, x, W, maxit, eps)


def defpcg(A: Operator, b, x, W, M: Operator, maxit: int = 0, eps: float = EPS):
    """defpcg(A, b, x, W, M; maxit=0) (defcg.jl:242-308) — note the reference's (A,b,x,W,M) order."""
    return _solve("defpcg", A, M, b, x, W, maxit, eps)


def eigcg(A: Operator, b, x, nvec: int, spdim: int, maxit: int = 0, eps: float = EPS):
    """eigcg(A, b, x, nvec, spdim; maxit=0) -> (x, it, res_norm, V[:, 1:nvec]) (eigcg.jl:27-123)."""
    return _solve("eigcg", A, None, b, x, None, maxit, eps, nvec, spdim)


def eigpcg(A: Operator, b, x, M: Operator, nvec: int, spdim: int, maxit: int = 0, eps: float = EPS):
    """eigpcg(A, b, x, M, nvec, spdim; maxit=0) -> (x, it, res_norm, V[:, 1:nvec]) (eigcg.jl:143-290)."""
    return _solve("eigpcg", A, M, b, x, None, maxit, eps, nvec, spdim)


def eigdefcg(A: Operator, b, x, W, spdim: int, maxit: int = 0, eps: float = EPS):
    """eigdefcg(A, b, x, W, spdim; maxit=0) -> (x, it, res_norm, V[:, 1:nvec]) (defcg.jl:111-223)."""
    return _solve("eigdefcg", A, None, b, x, W, maxit, eps, 0, spdim)


def eigdefpcg(A: Operator, b, x, M: Operator, W, spdim: int, maxit: int = 0, eps: float = EPS):
    """eigdefpcg(A, b, x, M, W, spdim; maxit=0) -> (x, it, res_norm, V[:, 1:nvec]) (defcg.jl:337-473) — (A,b,x,M,W) order."""
    return _solve("eigdefpcg", A, M, b, x, W, maxit, eps, 0, spdim)


def initcg(A: Operator, b, x, W, maxit: int = 0, eps: float = EPS):
    """initcg(A, b, x, W; maxit=0) (initcg.jl:28-75)."""
    return _solve("initcg", A, None, b, x, W, maxit, eps)


def initpcg(A: Operator, b, x, M: Operator, W, maxit: int = 0, eps: float = EPS):
    """initpcg(A, b, x, M, W; maxit=0) (initcg.jl:106-160)."""
    return _solve("initpcg", A, M, b, x, W, maxit, eps)


GlobalSchur.use_level_solver = _use_level_solver
