import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


_EIG_CACHE_MADE = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the x eigen-decomposition of an engine (two dgeev of 2048 x 2048 at 4097 points) is setup data that dozens of tests
    # rebuild for the same operator: keep it between the engines of one test session (csrc/hostmath.cc RPDE_EIG_CACHE;
    # child processes inherit the variable).  What is cached is what LAPACK returned the first time.
    global _EIG_CACHE_MADE
    if "RPDE_EIG_CACHE" not in os.environ:
        import tempfile
        _EIG_CACHE_MADE = tempfile.mkdtemp(prefix="rpde_eig_")
        os.environ["RPDE_EIG_CACHE"] = _EIG_CACHE_MADE


def pytest_unconfigure(config):
    if _EIG_CACHE_MADE:
        import shutil
        shutil.rmtree(_EIG_CACHE_MADE, ignore_errors=True)


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
