"""MI355X-native Schur-complement PCG hot path of venkovic/julia-phd-krylov-spdes.

The directory name is not a Python identifier; load it with `__graft_entry__.load_package()`,
which registers it as `krylov_spdes_amd`:

    fem  - host-side set-up mirror (mesh, partition, blocks, S_d, pinv) — feeds the path
    io   - the reference's npz mesh / partition / iteration-count files
    api  - reference-named operators / preconditioner / solvers over the C ABI (HIP only)
    _lib - ctypes binding of libmi355schur.so (include/mi355schur.h)
"""
from . import fem  # noqa: F401  (pure numpy/scipy; importable without the HIP library)

from . import io  # noqa: F401,E402

__all__ = ["fem", "io", "api", "_lib"]


def __getattr__(name):
    # api/_lib import lazily so that `fem` stays usable where the .so has not been built
    if name in ("api", "_lib"):
        import importlib
        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
