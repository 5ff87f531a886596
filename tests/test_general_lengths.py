"""Transform lengths other than Chebyshev n = 2^k + 1 / Fourier nx = 2^k: the reference accepts ANY n (funspace over rustdct /
rustfft / realfft) and its own benches and examples use such sizes -- Chebyshev n = 128, 264, 512, 1024
(benches/benchmark_navier.rs:6-7, benchmark_transform.rs:6: N = n - 1 = 127 and 263 are primes, 511 = 7 * 73, 1023 = 3 * 11 * 31),
periodic 18 x 13 (examples/navier_lnse_test_gradient.rs:11).  The engine runs them with Bluestein's algorithm inside the line
kernels (csrc/line_vm.h dct1_bluestein / rfft_bluestein, tables csrc/hostmath.cc bluestein_*_tables); lengths up to the limits
of the power-of-two plans (Chebyshev n <= 4097, Fourier nx <= 5461 for nx != 2^k).

CPU part: the host emulation build of the same kernel sources against the oracle (scipy.fft takes any n).  GPU part (-m gpu):
the HIP library through the C ABI, including the reference's criterion sizes at full size."""
import os
import subprocess
import sys

import numpy as np
import pytest

import rustpde_mpi_amd as R
from tests import checks as K
from tests.test_adjoint import check_lnse_gradient, check_lnse_parity, lnse_pair
from tests.test_sharded import _spawn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (axis-0 kind, n0, axis-1 kind, n1): primes, odd and even composites, the smallest sizes, the reference's bench sizes
SPACES_ANY = [("chebyshev", 12, "cheb_dirichlet", 13), ("cheb_dirichlet", 6, "cheb_neumann", 5),
              ("cheb_neumann", 128, "cheb_dirichlet", 264), ("cheb_dirichlet", 512, "cheb_neumann", 7),
              ("cheb_dirichlet", 10, "cheb_neumann", 300),
              ("fourier_r2c", 18, "cheb_dirichlet", 13), ("fourier_r2c", 2, "cheb_neumann", 9), ("fourier_r2c", 6, "cheb_neumann", 9),
              ("fourier_r2c", 17, "cheb_dirichlet", 13), ("fourier_r2c", 3, "cheb_dirichlet", 6), ("fourier_r2c", 100, "cheb_neumann", 9)]
# one case per kernel configuration of the line VM (128 / 256 / 512 / 1024 threads; Fourier: one slot of up to 17408 doubles)
SPACES_ANY_LONG = [("cheb_dirichlet", 6, "cheb_dirichlet", 1024), ("cheb_neumann", 1500, "cheb_dirichlet", 6),
                   ("cheb_neumann", 2048, "chebyshev", 5), ("cheb_neumann", 2050, "chebyshev", 5),
                   ("cheb_neumann", 3000, "cheb_neumann", 7), ("cheb_dirichlet", 6, "cheb_dirichlet", 4096),
                   ("fourier_r2c", 1000, "cheb_dirichlet", 5), ("fourier_r2c", 999, "cheb_dirichlet", 6),
                   ("fourier_r2c", 3000, "cheb_dirichlet", 5), ("fourier_r2c", 5461, "cheb_dirichlet", 5)]


@pytest.mark.parametrize("k0,n0,k1,n1", SPACES_ANY + SPACES_ANY_LONG)
def test_emu_space_ops(emu_lib, k0, n0, k1, n1):
    K.check_space_ops(emu_lib, k0, n0, k1, n1)


def test_emu_lengths_above_the_limits_are_refused(emu_lib):
    with pytest.raises(R.RpdeError, match="at most 4097"):
        R.Space2((K.KINDS["chebyshev"], 4098), (K.KINDS["chebyshev"], 9), library=emu_lib)
    with pytest.raises(R.RpdeError, match="5461"):
        R.Space2((K.KINDS["fourier_r2c"], 5462), (K.KINDS["chebyshev"], 9), library=emu_lib)
    with pytest.raises(R.RpdeError, match="16384"):
        R.Space2((K.KINDS["fourier_r2c"], 32768), (K.KINDS["chebyshev"], 9), library=emu_lib)


@pytest.mark.parametrize("k0,n0,k1,n1,c", [("cheb_dirichlet", 24, "cheb_dirichlet", 31, [1e-3, 2e-3]),
                                           ("cheb_neumann", 100, "cheb_neumann", 77, [1.0, 0.5]),
                                           ("fourier_r2c", 18, "cheb_dirichlet", 13, [1e-3, 1e-3]),
                                           ("fourier_r2c", 45, "cheb_neumann", 24, [1.0, 1.0])])
def test_emu_solvers(emu_lib, k0, n0, k1, n1, c):
    K.check_solvers(emu_lib, k0, n0, k1, n1, c, eig_mode="shared")   # the oracle's Poisson on the engine's eigenbasis: what is compared is the solve


@pytest.mark.parametrize("periodic,nx,ny,steps", [(False, 24, 25, 5), (False, 128, 128, 3), (False, 264, 265, 2), (False, 17, 2500, 2),
                                                  (True, 18, 13, 10), (True, 17, 13, 5), (True, 45, 24, 5), (True, 100, 31, 3)])
def test_emu_step(emu_lib, periodic, nx, ny, steps):
    """Navier2D::update at sizes of the reference's benches / examples (128, 264: benches/benchmark_navier.rs:6; 18 x 13:
    examples/navier_lnse_test_gradient.rs:11), odd Fourier lengths and a 2500-point Chebyshev line (1024-thread configuration)."""
    K.check_step_parity(emu_lib, periodic, nx, ny, 1e5 if not periodic else 1e4, 0.01, steps, check_at=[1, 2, steps])


def test_emu_step_hc(emu_lib):
    K.check_step_parity(emu_lib, False, 30, 31, 1e5, 0.01, 4, bc="hc")
    K.check_step_parity(emu_lib, True, 18, 21, 1e5, 0.01, 4, bc="hc")


def test_emu_bluestein_agrees_with_the_direct_transform():
    """The O(n^2) cosine sum (RPDE_DCT_DIRECT=1, n <= 500; until round 5 the only path for such lengths) is an independent
    evaluation of the same DCT-I: both meet the oracle in a child process that selects it."""
    code = ("from tests.emu.build_emu import build\nfrom rustpde_mpi_amd._capi import Lib\nfrom tests import checks as K\n"
            "lib = Lib(build())\nK.check_space_ops(lib, 'cheb_dirichlet', 10, 'cheb_neumann', 300)\n"
            "K.check_step_parity(lib, False, 12, 13, 1e5, 0.01, 2)\nprint('DIRECT-OK')\n")
    env = dict(os.environ, RPDE_DCT_DIRECT="1")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DIRECT-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_emu_lnse_at_the_size_of_the_reference_gradient_example(emu_lib, tmp_path):
    """examples/navier_lnse_test_gradient.rs runs periodic 18 x 13 (until round 5 the engine needed nx = 2^k and ran 16 x 13)."""
    check_lnse_parity(emu_lib, 18, 13, True, steps=4)
    check_lnse_parity(emu_lib, 18, 13, True, steps=4, adjoint=True)
    check_lnse_gradient(emu_lib, 18, 13, True, max_time=0.1, tmp_path=tmp_path)


def _engine_gradient_validation(lib, max_time, npts):
    """The reference's own validation of its adjoint gradient (examples/navier_lnse_test_gradient.rs: periodic 18 x 13, Ra = 3e3,
    Pr = 0.1, dt = 0.01, amplitude 1e-3; |g_fd - g_adj| / |g_adj| accepted at 0.3) on the ENGINE, sub-sampled."""
    nav, ora = lnse_pair(lib, 18, 13, True, 3e3, 0.1, 0.01)
    base = {k: getattr(nav, k).vhat.copy() for k in ("velx", "vely", "temp")}
    _, g_adj = nav.grad_adjoint(max_time, None, 0.5, 0.5, None, filename=None)
    for k in base:
        getattr(nav, k).vhat = base[k]
    rng = np.random.default_rng(11)
    pts = [(k, int(rng.integers(18)), int(rng.integers(2, 11))) for k in ("velx", "vely", "temp") for _ in range(npts)]
    g_fd = nav.grad_fd(max_time, None, 0.5, 0.5, points=pts, filename=None)
    names = ("velx", "vely", "temp")
    ga = np.array([-dict(zip(names, g_adj))[k][i, j] for k, i, j in pts])
    gf = np.array([dict(zip(names, g_fd))[k][i, j] for k, i, j in pts])
    return float(np.linalg.norm(ga - gf) / np.linalg.norm(ga))


def test_emu_engine_passes_the_reference_gradient_validation(emu_lib):
    assert _engine_gradient_validation(emu_lib, 0.3, 6) < 0.3   # (the GPU test runs horizon 1)


CASES_SHARDED = [(False, 24, 25, 1e5, 0.01, 3, 1.0), (True, 18, 13, 1e4, 0.01, 4, 1.0), (False, 70, 131, 1e5, 0.01, 3, 1.0, "rbc", "single"),   # against the one-rank engine: same eigenbasis
                 (True, 45, 24, 1e4, 0.01, 3, 1.0), (False, 30, 31, 1e5, 0.01, 3, 1.0, "hc")]


@pytest.mark.parametrize("world", [2, 3])
def test_emu_sharded(world, tmp_path, emu_lib):
    """The pencil-sharded engine (Navier2DMpi) at such sizes: ragged partitions of 24 / 25 / 70 / 131 rows over 2 and 3 ranks."""
    res = _spawn(world, emu_lib.path, False, CASES_SHARDED, tmp_path)
    assert len(res) == len(CASES_SHARDED)
    for r in res:
        for k, e in r["err"].items():
            assert e < 1e-10, (r["case"], k, e)


# ------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("k0,n0,k1,n1", SPACES_ANY + SPACES_ANY_LONG + [("cheb_dirichlet", 512, "cheb_dirichlet", 512),
                                                                      ("cheb_neumann", 1024, "cheb_dirichlet", 1024),
                                                                      ("cheb_dirichlet", 264, "cheb_neumann", 265),
                                                                      ("chebyshev", 3001, "chebyshev", 2999),
                                                                      ("fourier_r2c", 1536, "cheb_dirichlet", 769),
                                                                      ("fourier_r2c", 5000, "cheb_neumann", 4000)])
def test_gpu_space_ops(hip_lib, k0, n0, k1, n1):
    K.check_space_ops(hip_lib, k0, n0, k1, n1)


@pytest.mark.gpu
@pytest.mark.parametrize("k0,n0,k1,n1,c", [("cheb_neumann", 100, "cheb_neumann", 77, [1.0, 0.5]),
                                           ("cheb_neumann", 512, "cheb_neumann", 264, [1.0, 1.0]),
                                           ("fourier_r2c", 18, "cheb_dirichlet", 13, [1e-3, 1e-3]),
                                           ("fourier_r2c", 600, "cheb_neumann", 300, [1.0, 1.0])])
def test_gpu_solvers(hip_lib, k0, n0, k1, n1, c):
    K.check_solvers(hip_lib, k0, n0, k1, n1, c, eig_mode="shared")


@pytest.mark.gpu
@pytest.mark.parametrize("nx,ny,steps", [(24, 25, 10), (128, 128, 20), (264, 264, 10), (264, 265, 10), (512, 512, 10)])
def test_gpu_confined_step_at_the_reference_bench_sizes(hip_lib, nx, ny, steps):
    """benches/benchmark_navier.rs:6-7, 30-37: Navier2D::new_confined(n, n, 1e5, 1., 0.01, 1., "rbc") for n = 128, 264, 512."""
    K.check_step_parity(hip_lib, False, nx, ny, 1e5, 0.01, steps, check_at=[1, 2, steps])


@pytest.mark.gpu
def test_gpu_confined_step_1024(hip_lib):
    """benches/benchmark_transform.rs:6 goes to n = 1024; a confined step there (N = 1023 = 3 * 11 * 31: M = 2048), in a child process."""
    K.run_isolated("check_step_parity(lib, False, 1024, 1024, 1e7, 1e-3, 5, check_at=[1, 5])")


@pytest.mark.gpu
@pytest.mark.parametrize("nx,ny", [(2500, 40), (40, 3000), (1300, 1200)])
def test_gpu_confined_step_long_lines(hip_lib, nx, ny):
    """Chebyshev lines of 1026 .. 4096 points: M = 4096 on 512 threads, M = 8192 on 1024 threads (two 70 KB slots)."""
    K.check_step_parity(hip_lib, False, nx, ny, 1e6, 1e-3, 2, check_at=[2], tol=1e-9 if min(nx, ny) < 100 else 1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("nx,ny,steps", [(18, 13, 20), (17, 13, 10), (45, 24, 10), (100, 31, 10), (600, 257, 5), (1000, 400, 3), (3000, 129, 3)])
def test_gpu_periodic_step(hip_lib, nx, ny, steps):
    K.check_step_parity(hip_lib, True, nx, ny, 1e5, 0.005, steps, check_at=[1, 2, steps])


@pytest.mark.gpu
def test_gpu_step_hc(hip_lib):
    K.check_step_parity(hip_lib, False, 264, 200, 1e5, 0.01, 4, bc="hc")
    K.check_step_parity(hip_lib, True, 90, 101, 1e5, 0.01, 4, bc="hc")


@pytest.mark.gpu
def test_gpu_lnse_at_the_size_of_the_reference_gradient_example(hip_lib, tmp_path):
    check_lnse_parity(hip_lib, 18, 13, True, steps=4)
    check_lnse_parity(hip_lib, 18, 13, True, steps=4, adjoint=True)
    check_lnse_gradient(hip_lib, 18, 13, True, max_time=0.2, tmp_path=tmp_path)


@pytest.mark.gpu
def test_gpu_engine_passes_the_reference_gradient_validation(hip_lib):
    """examples/navier_lnse_test_gradient.rs on the engine at ITS size (18 x 13), horizon 1, 5 points per field."""
    assert _engine_gradient_validation(hip_lib, 1.0, 5) < 0.3


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2])
def test_gpu_sharded(world, tmp_path, hip_lib):
    res = _spawn(world, hip_lib.path, True, CASES_SHARDED + [(False, 512, 512, 1e5, 0.01, 2, 1.0, "rbc", "single")], tmp_path)
    for r in res:
        for k, e in r["err"].items():
            assert e < 1e-10, (r["case"], k, e)
