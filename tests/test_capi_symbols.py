"""The C-ABI library loads and exports every symbol include/rustpde_hip.h declares (no compute)."""
import ctypes
import os
import re

import pytest

import rustpde_mpi_amd as R
from rustpde_mpi_amd._capi import SIGNATURES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "rustpde_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rpde_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert header_symbols() == sorted(SIGNATURES)


def test_hip_library_exports_every_symbol():
    if not os.path.exists(R.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    dll = ctypes.CDLL(R.LIB_PATH)
    for name in header_symbols():
        assert hasattr(dll, name), name
    dll.rpde_version.restype = ctypes.c_char_p
    assert b"HIP gfx950" in dll.rpde_version()
    assert dll.rpde_is_device_build() == 1


def test_missing_library_fails_loudly(tmp_path):
    from rustpde_mpi_amd._capi import Lib, RpdeError
    with pytest.raises(RpdeError, match="no CPU fallback"):
        Lib(str(tmp_path / "librustpde_hip.so"))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "rustpde_mpi_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cc", ".h")):
                text = open(os.path.join(root, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "librustpde_emu" not in text, f
