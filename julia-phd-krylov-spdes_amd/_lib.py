"""ctypes binding of libmi355schur.so — the C ABI declared in include/mi355schur.h.

There is no CPU fallback: if the shared library is missing this module raises at import of
the symbols, and if no gfx950 device is visible `mi_ctx_create` fails with MI_ERR_NO_DEVICE.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI355SCHUR_LIB") or os.path.join(_PKG, "libmi355schur.so")   # override: A/B builds
CSRC = os.path.join(_PKG, "csrc")

MI_OK = 0
MI_ERR_BAD_ARG, MI_ERR_HIP, MI_ERR_SINGULAR, MI_ERR_RES_CAPACITY = -1, -2, -3, -4
MI_ERR_COMM, MI_ERR_NO_DEVICE, MI_ERR_CALLBACK, MI_ERR_BOUNDS = -5, -6, -7, -8
MI_PTR_HOST, MI_PTR_DEVICE = 0, 1
MI_COMM_ID_BYTES = 128
MI_PEER_HANDLE_BYTES = 64

i64 = C.c_int64
i64p = C.POINTER(C.c_int64)
i64pp = C.POINTER(i64p)
f64p = C.POINTER(C.c_double)
f64pp = C.POINTER(f64p)
vp = C.c_void_p
INTERIOR_SOLVE_FN = C.CFUNCTYPE(C.c_int, vp, i64, i64, f64p, f64p)

# name -> argtypes (restype is int unless listed in _RESTYPE)
SIGNATURES = {
    "mi_version": [],
    "mi_last_error": [],
    "mi_device_count": [C.POINTER(C.c_int)],
    "mi_ctx_create": [C.c_int, C.POINTER(vp)],
    "mi_ctx_destroy": [vp],
    "mi_ctx_set_pointer_mode": [vp, C.c_int],
    "mi_ctx_set_stream": [vp, vp],
    "mi_ctx_get_stream": [vp, C.POINTER(vp)],
    "mi_ctx_synchronize": [vp],
    "mi_ctx_set_chunk": [vp, C.c_int],
    "mi_comm_unique_id": [vp],
    "mi_ctx_comm_init": [vp, vp, C.c_int, C.c_int],
    "mi_ctx_comm_destroy": [vp],
    "mi_ctx_allreduce_sum": [vp, vp, i64],
    "mi_csr_create": [vp, i64, i64, i64p, i64p, f64p, C.c_int, C.POINTER(vp)],
    "mi_diag_create": [vp, i64, f64p, C.POINTER(vp)],
    "mi_schur_assembled_create": [vp, i64, i64, i64p, i64pp, f64pp, C.c_int, i64, i64, C.POINTER(vp)],
    "mi_nn_create": [vp, i64, i64, i64p, i64pp, f64pp, i64p, C.c_int, i64, i64, C.POINTER(vp)],
    "mi_schur_matfree_create": [vp, i64, i64, i64p, i64p, i64pp, i64pp, i64pp, f64pp, i64pp, i64pp, f64pp,
                                INTERIOR_SOLVE_FN, vp, C.c_int, i64, i64, C.POINTER(vp)],
    "mi_schur_matfree_device_create": [vp, i64, i64, i64p, i64p, i64pp, i64pp, i64pp, f64pp, i64pp, i64pp, f64pp,
                                       i64pp, i64pp, f64pp, C.c_double, C.c_int, i64, i64, C.POINTER(vp)],
    "mi_schur_global_create": [vp, i64, i64, i64p, i64pp, i64pp, f64pp, i64p, i64p, f64p,
                               INTERIOR_SOLVE_FN, vp, C.c_int, C.POINTER(vp)],
    "mi_schur_global_device_create": [vp, i64, i64, i64p, i64pp, i64pp, f64pp, i64pp, i64pp, f64pp, i64p, i64p, f64p,
                                      C.c_double, C.c_int, C.POINTER(vp)],
    "mi_op_size": [vp, i64p],
    "mi_op_apply": [vp, vp, vp],
    "mi_op_bytes": [vp, i64p, i64p],
    "mi_op_apply_dominant": [vp, vp, C.c_int],
    "mi_op_time_dominant": [vp, vp, C.c_int, f64p],
    "mi_op_destroy": [vp],
    "mi_dot": [vp, i64, vp, vp, f64p],
    "mi_norm2": [vp, i64, vp, f64p],
    "mi_axpy": [vp, i64, C.c_double, vp, vp],
    "mi_axpby": [vp, i64, C.c_double, vp, C.c_double, vp],
    "mi_cg": [vp, vp, vp, i64, C.c_double, f64p, i64, i64p],
    "mi_pcg": [vp, vp, vp, vp, i64, C.c_double, f64p, i64, i64p],
    "mi_defcg": [vp, vp, vp, vp, i64, i64, C.c_double, f64p, i64, i64p],
    "mi_defpcg": [vp, vp, vp, vp, vp, i64, i64, C.c_double, f64p, i64, i64p],
    "mi_loopback_group_create": [C.c_int, C.POINTER(vp)],
    "mi_loopback_group_destroy": [vp],
    "mi_ctx_loopback_init": [vp, vp, C.c_int],
    "mi_loopback_group_set_mode": [vp, C.c_int],
    "mi_ctx_peer_init": [vp, C.c_int, C.c_int, i64],
    "mi_ctx_peer_export": [vp, vp, C.POINTER(vp)],
    "mi_ctx_peer_import": [vp, C.c_int, vp, vp],
    "mi_ctx_peer_ready": [vp],
    "mi_ctx_set_exchange": [vp, C.c_int],
    "mi_ctx_query": [vp, C.c_int, i64p],
    "mi_schur_setup_keep_levels": [vp, C.c_int],
    "mi_schur_setup_interior_solve": [vp, vp, vp],
    "mi_schur_matfree_interior_levels": [vp, vp],
    "mi_assembly_plan_create": [vp, i64, i64, i64p, C.c_int, f64p, f64p, f64p, f64p, i64, i64, i64p, i64p, C.POINTER(vp)],
    "mi_assembly_run": [vp, vp, vp],
    "mi_assembly_plan_destroy": [vp],
    "mi_schur_matfree_set_values": [vp, vp, vp, vp],
    "mi_schur_setup_create": [vp, i64, i64p, i64p, i64pp, i64pp, i64pp, i64pp, i64pp, i64pp, C.c_int, C.POINTER(vp)],
    "mi_schur_setup_run": [vp, vp, vp, vp, vp, vp, vp],
    "mi_schur_setup_destroy": [vp],
    "mi_nn_pinv": [vp, i64, i64p, vp, C.c_double, vp],
    "mi_dense_set_blocks": [vp, vp],
    "mi_schur_matfree_rhs": [vp, vp, vp, vp],
    "mi_schur_matfree_interior_solutions": [vp, vp, vp, vp],
    "mi_schur_interior_precond": [vp, C.c_int],
    "mi_schur_interior_iterations": [vp, i64p],
    "mi_eigcg": [vp, vp, vp, i64, i64, i64, C.c_double, f64p, i64, i64p, vp],
    "mi_eigpcg": [vp, vp, vp, vp, i64, i64, i64, C.c_double, f64p, i64, i64p, vp],
    "mi_eigdefcg": [vp, vp, vp, vp, i64, i64, i64, C.c_double, f64p, i64, i64p, vp],
    "mi_eigdefpcg": [vp, vp, vp, vp, vp, i64, i64, i64, C.c_double, f64p, i64, i64p, vp],
    "mi_initcg": [vp, vp, vp, vp, i64, i64, C.c_double, f64p, i64, i64p],
    "mi_initpcg": [vp, vp, vp, vp, vp, i64, i64, C.c_double, f64p, i64, i64p],
    "mi_event_create": [C.POINTER(vp)],
    "mi_event_record": [vp, vp],
    "mi_event_elapsed_ms": [vp, vp, f64p],
    "mi_event_destroy": [vp],
}
_RESTYPE = {"mi_last_error": C.c_char_p}

_lib = None


def build(force: bool = False) -> str:
    """Compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-s", "-C", CSRC]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C {CSRC}` (or __graft_entry__.build()). "
            "There is no CPU fallback for the MI355X hot path.")
    if not os.environ.get("MI355_NO_TORCH"):
        # torch ships its own libamdhip64.so.7 / librccl.so.1; import it first so that the whole
        # process (torch tensors, our kernels, RCCL) shares ONE HIP runtime, resolved by SONAME.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(L, name)          # AttributeError here = header and library disagree
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, C.c_int)
    _lib = L
    return L


class MiError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libmi355schur error {code}: {msg}")
        self.code = code


class SingularException(MiError, ArithmeticError):
    """LinearAlgebra.SingularException from `WtAW \\ mu` (defcg.jl:53, 273)."""


class BoundsError(MiError, IndexError):
    """`res_norm[it]` beyond its n entries (cg.jl:23,47); eigCG family: `V[:, nev+1]` / `eigvecs(...)[:, 1:nvec]` out of range."""


def check(rc: int) -> None:
    if rc == MI_OK:
        return
    msg = load().mi_last_error().decode("utf-8", "replace")
    if rc == MI_ERR_SINGULAR:
        raise SingularException(rc, msg)
    if rc in (MI_ERR_RES_CAPACITY, MI_ERR_BOUNDS):
        raise BoundsError(rc, msg)
    raise MiError(rc, msg)
