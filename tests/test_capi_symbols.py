"""The C-ABI shared library loads and exports every entry point that include/fastdepth_hip.h declares
(no compute calls: there is no GPU in the CPU test tier)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "fastdepth_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fd_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_boundary():
    names = declared_functions()
    for must in ("fd_plan_create", "fd_plan_pack_weights", "fd_forward", "fd_plan_destroy", "fd_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from fastdepth_hip import capi
    if not os.path.exists(capi.DEFAULT_LIB):
        import importlib.util
        spec = importlib.util.spec_from_file_location("fd_build", os.path.join(REPO, "fast-depth_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod); mod.build()
    lib = ctypes.CDLL(capi.DEFAULT_LIB)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(capi.EXPORTS) == declared_functions()       # the Python binding covers the whole header
    assert b"gfx950" in ctypes.cast(lib.fd_version, ctypes.CFUNCTYPE(ctypes.c_char_p))()


def test_plan_api_without_a_gpu():
    """Plan construction and its error paths are host-only code: they work on the real library with no device."""
    from fastdepth_hip import capi
    lib = capi.load()
    h = ctypes.c_void_p()
    bad = (capi.LayerDesc * 1)(capi.LayerDesc(capi.FD_OP_PW, 3, 1, 1, 1, 1, -1, 0, -1, 0))
    assert lib.fd_plan_create(bad, 1, 1, 224, 224, capi.FD_F32, 0, ctypes.byref(h)) == -1
    assert b"pointwise" in lib.fd_last_error()
    assert lib.fd_plan_create(bad, 1, 1, 228, 304, capi.FD_F32, 0, ctypes.byref(h)) == -1
    assert b"multiples of 32" in lib.fd_last_error()


def test_missing_library_fails_loudly(tmp_path):
    from fastdepth_hip import capi
    with pytest.raises(capi.FastDepthError):
        capi.load(str(tmp_path / "libfastdepth_hip.so"))
