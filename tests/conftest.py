import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def emu_lib():
    """Host emulation build of the kernel sources (tests/emu/README.md) -- never the product."""
    from tests.emu.build_emu import build
    from rustpde_mpi_amd._capi import Lib
    lib = Lib(build())
    assert not lib.is_device_build
    return lib


@pytest.fixture(scope="session")
def hip_lib():
    """The product library, through the same loader the package uses (fails loudly if missing)."""
    import rustpde_mpi_amd as R
    lib = R.lib()
    assert lib.is_device_build
    return lib
