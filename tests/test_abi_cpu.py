"""CPU suite: the C-ABI library loads, exports every symbol include/mi355schur.h declares, and fails
loudly (no CPU fallback) where there is no GPU. No compute call is made here."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mi355schur.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(?:int|const char \*)\s*(mi_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_and_binding_agree(pkg):
    declared = _declared_symbols()
    assert len(declared) >= 35
    assert set(declared) == set(pkg._lib.SIGNATURES)        # the ctypes table covers exactly the header


def test_library_exports_every_declared_symbol(pkg):
    import __graft_entry__ as graft
    graft.build()
    L = C.CDLL(pkg._lib.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(L, name), f"{name} declared in mi355schur.h but not exported"
    assert pkg._lib.load().mi_version() >= 100


def test_no_torch_types_or_cxx_in_header():
    text = open(os.path.join(ROOT, "include", "mi355schur.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)       # declarations only
    for bad in ("torch", "at::", "std::", "template", "class ", "&"):
        assert bad not in text


def _has_gpu(pkg):
    n = C.c_int(0)
    pkg._lib.load().mi_device_count(C.byref(n))
    return n.value > 0


def test_fails_loudly_without_gpu(pkg):
    if _has_gpu(pkg):
        pytest.skip("a GPU is visible; covered by the gpu suite")
    with pytest.raises(pkg._lib.MiError) as e:
        pkg.api.Context(0)
    assert e.value.code == pkg._lib.MI_ERR_NO_DEVICE and "no CPU fallback" in str(e.value)


def test_null_arguments_return_bad_arg(pkg):
    L = pkg._lib.load()
    BAD = pkg._lib.MI_ERR_BAD_ARG
    assert L.mi_ctx_create(0, None) == BAD
    assert L.mi_ctx_set_pointer_mode(None, 0) == BAD
    assert L.mi_op_apply(None, None, None) == BAD
    assert L.mi_op_size(None, None) == BAD
    it = C.c_int64()
    assert L.mi_cg(None, None, None, 0, 1e-7, None, 0, C.byref(it)) == BAD
    assert L.mi_pcg(None, None, None, None, 0, 1e-7, None, 0, C.byref(it)) == BAD
    assert L.mi_defpcg(None, None, None, None, None, 0, 0, 1e-7, None, 0, C.byref(it)) == BAD
    assert L.mi_dot(None, 3, None, None, None) == BAD
    assert b"NULL" in L.mi_last_error() or b"bad" in L.mi_last_error()
    assert L.mi_ctx_destroy(None) == 0 and L.mi_op_destroy(None) == 0   # destroying NULL is a no-op


def test_product_never_imports_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg_dir = os.path.join(ROOT, "julia-phd-krylov-spdes_amd")
    for base, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")) or f == "Makefile":
                src = open(os.path.join(base, f), errors="replace").read()
                for bad in ("import oracle", "from oracle", "krylov_oracle", "oracle/", "orc_"):
                    assert bad not in src, f"{f} reaches into the oracle ({bad})"


def test_shard_domains_covers_all_subdomains(pkg):
    for ndom in (1, 4, 8, 13):
        for nr in (1, 2, 4, 8):
            parts = [pkg.api.shard_domains(ndom, r, nr) for r in range(nr)]
            assert parts[0][0] == 0 and parts[-1][1] == ndom
            assert all(parts[i][1] == parts[i + 1][0] for i in range(nr - 1))
    assert [pkg.api.shard_domains(8, r, 4) for r in range(4)] == [(0, 2), (2, 4), (4, 6), (6, 8)]


def test_plain_c_driver_compiles_against_the_header(pkg, tmp_path):
    """tests/c/abi_drive.c (run by the gpu suite) must compile as C11 with -Werror against include/mi355schur.h and link
    against the library: the header is then a valid C header, not only something ctypes mirrors."""
    import subprocess
    import __graft_entry__ as graft
    graft.build()
    lib_dir = os.path.dirname(pkg._lib.LIB_PATH)
    exe = str(tmp_path / "abi_drive")
    subprocess.check_call(["gcc", "-O1", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_drive.c"), "-o", exe, "-L", lib_dir, "-lmi355schur", "-lm",
                           f"-Wl,-rpath,{lib_dir}"])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr       # no GPU work without arguments
