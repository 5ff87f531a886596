"""CPU-side checks of the kernel sources through their host emulation build (tests/emu/README.md):
line programs, tables, the fused step schedule -- against the oracle.  The GPU parity tests proper
are in test_gpu_parity.py and run the same checks on the HIP library."""
import os

import numpy as np
import pytest

import rustpde_mpi_amd as R
from tests import checks as K

SPACES = [("cheb_dirichlet", 7, "cheb_dirichlet", 7), ("cheb_neumann", 17, "cheb_dirichlet", 33),
          ("chebyshev", 33, "chebyshev", 17), ("cheb_dirichlet", 129, "cheb_neumann", 65),
          ("fourier_r2c", 16, "cheb_dirichlet", 9), ("fourier_r2c", 64, "cheb_neumann", 33),
          ("cheb_dirichlet", 10, "cheb_neumann", 300),   # direct (non power-of-two) transform
          ("cheb_neumann", 9, "cheb_dirichlet", 1025), ("cheb_dirichlet", 2049, "cheb_dirichlet", 9),
          ("cheb_dirichlet", 4097, "cheb_neumann", 9), ("fourier_r2c", 4096, "cheb_dirichlet", 9)]


@pytest.mark.parametrize("k0,n0,k1,n1", SPACES)
def test_space_ops(emu_lib, k0, n0, k1, n1):
    K.check_space_ops(emu_lib, k0, n0, k1, n1)


@pytest.mark.parametrize("k0,n0,k1,n1,c", [
    ("cheb_dirichlet", 33, "cheb_dirichlet", 17, [1e-3, 2e-3]),
    ("cheb_neumann", 65, "cheb_dirichlet", 129, [3e-5, 1e-5]),
    ("cheb_neumann", 129, "cheb_neumann", 65, [1.0, 0.5]),
    ("fourier_r2c", 64, "cheb_dirichlet", 33, [1e-3, 1e-3]),
    ("fourier_r2c", 32, "cheb_neumann", 65, [1.0, 1.0])])
def test_solvers(emu_lib, k0, n0, k1, n1, c):
    K.check_solvers(emu_lib, k0, n0, k1, n1, c)


def test_reference_known_answers(emu_lib):
    K.check_reference_known_answers(emu_lib)


@pytest.mark.parametrize("n", [257, 1025, 2049, 4097])
def test_dct_line_backward(emu_lib, n):
    K.check_dct_line_backward(emu_lib, n)


def test_whole_line_kernel_rejects_other_lengths(emu_lib):
    a = np.zeros((1, 129)); out = np.zeros((1, 129))
    with pytest.raises(R.RpdeError, match="not covered"):
        emu_lib.call("rpde_dct_line_backward", 0, 129, R._capi.ptr(a), 1, R._capi.ptr(out), 0)


@pytest.mark.parametrize("nx,ny,ra,dt,steps", [(17, 17, 1e4, 0.01, 5), (33, 33, 1e5, 0.01, 20),
                                               (65, 33, 1e5, 0.01, 10), (33, 65, 1e5, 0.01, 10),
                                               (17, 257, 1e5, 0.01, 4)])   # ny = 257: S2 runs the whole-line kernel
def test_confined_step(emu_lib, nx, ny, ra, dt, steps):
    K.check_step_parity(emu_lib, False, nx, ny, ra, dt, steps, check_at=[1, 2, steps])


def _has_line_program(nav, tag):
    kinds = [kind for t, _, _, _, kind in nav.schedule() if t.startswith(tag) or tag in t]
    assert kinds, tag
    return all(k == "line program" for k in kinds)


def test_confined_step_s1_through_the_whole_line_kernel(emu_lib, monkeypatch):
    """Value and x-derivative of the state lines (Dirichlet stencil for u, v; Neumann table for T; suffix-sum
    derivative) through csrc/dct_line.h -- nx = 257 is a length the emulation build covers; RPDE_S1_LINE=0 keeps the
    line program (A/B switch)."""
    K.check_step_parity(emu_lib, False, 257, 17, 1e5, 0.01, 4, check_at=[1, 4])
    nav, _ = K.make_pair(emu_lib, False, 257, 17, 1e5, 1.0, 0.01, 1.0)
    assert not _has_line_program(nav, "S1 x")
    monkeypatch.setenv("RPDE_S1_LINE", "0")
    K.check_step_parity(emu_lib, False, 257, 17, 1e5, 0.01, 2)
    nav, _ = K.make_pair(emu_lib, False, 257, 17, 1e5, 1.0, 0.01, 1.0)
    assert _has_line_program(nav, "S1 x")


@pytest.mark.parametrize("nx,ny,eig", [(257, 17, "parity"), (257, 33, "parity"), (1025, 17, "shared")])
def test_confined_step_s8_through_the_whole_line_kernel(emu_lib, monkeypatch, nx, ny, eig):
    """x part of the velocity correction (csrc/corr_line.h: stencil, derivative and projection folded into three taps +
    two sweeps + a rank-one term per line) against the oracle, and against the line program of the stage
    (RPDE_S8_LINE=0: the A/B switch)."""
    K.check_step_parity(emu_lib, False, nx, ny, 1e5, 0.01, 3, check_at=[1, 3], eig_mode=eig)
    nav, _ = K.make_pair(emu_lib, False, nx, ny, 1e5, 1.0, 0.01, 1.0)
    assert not _has_line_program(nav, "S8 x")
    nav.update(2)
    monkeypatch.setenv("RPDE_S8_LINE", "0")
    ref, _ = K.make_pair(emu_lib, False, nx, ny, 1e5, 1.0, 0.01, 1.0)
    assert _has_line_program(ref, "S8 x")
    ref.update(2)
    for k in ("velx", "vely", "temp", "pres"):
        assert K.rel(getattr(nav, k).vhat, getattr(ref, k).vhat) < (1e-10 if k == "pres" else 1e-11), k   # two orders of summation


@pytest.mark.parametrize("periodic,nx,ny", [(False, 33, 257), (False, 129, 129), (False, 17, 1025), (True, 32, 513)])
def test_column_scans_in_one_pass(emu_lib, monkeypatch, periodic, nx, ny):
    """The single-pass column scans of one rank (csrc/colscan1.h: super-blocks of W blocks per workgroup, aggregates, one
    meeting of a column tile's workgroups, correction of the zero-inflow rows) against the oracle and against the three
    kernels of colscan.h (RPDE_COL_ONEPASS=0: the A/B switch) -- ny = 257 / 1025: 8 / 32 blocks, several super-blocks."""
    K.check_step_parity(emu_lib, periodic, nx, ny, 1e5, 0.01, 3, check_at=[1, 3])
    nav, _ = K.make_pair(emu_lib, periodic, nx, ny, 1e5, 1.0, 0.01, 1.0)
    scans = ("hholtz-y (column scan)", "correction-y (column scan)")    # the periodic step has C4 only
    kinds = {t: kind for t, _, _, _, kind in nav.schedule() if any(c in t for c in scans)}
    assert kinds and all(k == "column scan (one pass)" for k in kinds.values()), kinds
    nav.update(3)
    monkeypatch.setenv("RPDE_COL_ONEPASS", "0")
    ref, _ = K.make_pair(emu_lib, periodic, nx, ny, 1e5, 1.0, 0.01, 1.0)
    assert all(kind == "column scan" for t, _, _, _, kind in ref.schedule() if any(c in t for c in scans))
    ref.update(3)
    for k in ("velx", "vely", "temp", "pres"):
        assert K.rel(getattr(nav, k).vhat, getattr(ref, k).vhat) < (1e-10 if k == "pres" else 1e-12), k


@pytest.mark.parametrize("nx,ny,bc", [(256, 17, "rbc"), (256, 33, "rbc"), (1024, 17, "rbc"), (256, 33, "hc")])
def test_periodic_step_fourier_lines_through_the_whole_line_kernels(emu_lib, monkeypatch, nx, ny, bc):
    """S1 (spectral line -> physical values and x-derivative, two inverse real FFTs per workgroup) and S3 (forward real FFT,
    2/3 rule, right-hand side, diagonal Helmholtz factor) of the periodic step as whole-line kernels (csrc/rfft_line.h)
    against the oracle, and against the line programs of the stages (RPDE_S1_LINE=0 / RPDE_S3_LINE=0: the A/B switches)."""
    K.check_step_parity(emu_lib, True, nx, ny, 1e5, 0.01, 3, check_at=[1, 3], bc=bc)
    nav, _ = K.make_pair(emu_lib, True, nx, ny, 1e5, 1.0, 0.01, 1.0, bc=bc)
    kinds = {t: kind for t, _, _, _, kind in nav.schedule()}
    assert kinds["S1 x: state -> phys-x + d/dx"].startswith("whole-line transform pair"), kinds
    assert all(kinds[f"S3 x: rhs + hholtz-x {f}"].startswith("whole-line rhs") for f in ("velx", "vely", "temp")), kinds
    nav.update(3)
    monkeypatch.setenv("RPDE_S1_LINE", "0")
    monkeypatch.setenv("RPDE_S3_LINE", "0")
    ref, _ = K.make_pair(emu_lib, True, nx, ny, 1e5, 1.0, 0.01, 1.0, bc=bc)
    assert _has_line_program(ref, "S1 x") and _has_line_program(ref, "S3 x")
    ref.update(3)
    for k in ("velx", "vely", "temp", "pres"):
        assert K.rel(getattr(nav, k).vhat, getattr(ref, k).vhat) < (1e-10 if k == "pres" else 1e-11), k


@pytest.mark.parametrize("nx,ny,eig", [(257, 17, "parity"), (257, 33, "parity"), (1025, 17, "shared")])
def test_confined_step_s5_through_the_whole_line_kernel(emu_lib, monkeypatch, nx, ny, eig):
    """Divergence + x preconditioner of the Poisson solve (csrc/div_line.h) against the oracle, and against the line program
    of the stage (RPDE_S5_LINE=0: the A/B switch); the divergence norm is what Integrate::exit looks at."""
    K.check_step_parity(emu_lib, False, nx, ny, 1e5, 0.01, 3, check_at=[1, 3], eig_mode=eig)
    nav, _ = K.make_pair(emu_lib, False, nx, ny, 1e5, 1.0, 0.01, 1.0)
    assert not _has_line_program(nav, "S5 x")
    nav.update(2)
    monkeypatch.setenv("RPDE_S5_LINE", "0")
    ref, _ = K.make_pair(emu_lib, False, nx, ny, 1e5, 1.0, 0.01, 1.0)
    assert _has_line_program(ref, "S5 x")
    ref.update(2)
    for k in ("velx", "vely", "temp", "pres"):
        assert K.rel(getattr(nav, k).vhat, getattr(ref, k).vhat) < (1e-10 if k == "pres" else 1e-11), k
    assert abs(nav.div_norm() - ref.div_norm()) < 1e-9 * max(1.0, ref.div_norm())


def test_s6_poisson_rows_as_one_kernel(emu_lib, monkeypatch):
    """S6 of the confined step -- y preconditioner and one factorised banded solve per eigen row of the Poisson problem -- as
    the whole-line kernel csrc/prow_line.h (the row's factors in a second, 16-element chunk-major copy: PoissonOp::rows16):
    against the oracle (257-point y-lines, and 1025: one wave per line) and against the line program of the stage
    (RPDE_S6_LINE=0: the A/B switch)."""
    K.check_step_parity(emu_lib, False, 33, 257, 1e6, 2e-3, 3, check_at=[1, 3])
    K.check_step_parity(emu_lib, False, 17, 1025, 1e6, 2e-3, 2, check_at=[2])
    nav, ora = K.make_pair(emu_lib, False, 33, 257, 1e6, 1.0, 2e-3, 1.0)
    assert {t: kind for t, _, _, _, kind in nav.schedule()}["S6 y: poisson rows"] == "whole-line poisson rows"
    nav.update(3)
    monkeypatch.setenv("RPDE_S6_LINE", "0")
    nav0, _ = K.make_pair(emu_lib, False, 33, 257, 1e6, 1.0, 2e-3, 1.0)
    assert _has_line_program(nav0, "S6 y")
    nav0.update(3)
    f1, f0 = nav.physical_fields(), nav0.physical_fields()
    for k in f0:
        assert K.rel(f1[k], f0[k]) < 1e-12, (k, K.rel(f1[k], f0[k]))
    # RPDE_S6_KEEP=0: the factors of a row read twice instead of kept in registers -- the same arithmetic
    monkeypatch.delenv("RPDE_S6_LINE")
    monkeypatch.setenv("RPDE_S6_KEEP", "0")
    nav2, _ = K.make_pair(emu_lib, False, 33, 257, 1e6, 1.0, 2e-3, 1.0)
    nav2.update(3)
    f2 = nav2.physical_fields()
    for k in f1:
        assert np.array_equal(f1[k], f2[k]), k
    # RPDE_S6_DERIVE=1 (round 6, A/B): p2 alone is read per eigen row, q1 / q2 / r2 are derived from it in the kernel -- the
    # factors agree with the tabulated ones to their last bit but one, the fields to round-off; half the bytes of the stage
    monkeypatch.delenv("RPDE_S6_KEEP")
    monkeypatch.setenv("RPDE_S6_DERIVE", "1")
    nav3, _ = K.make_pair(emu_lib, False, 33, 257, 1e6, 1.0, 2e-3, 1.0)
    assert 2 * {t: b for t, b, _, _, _ in nav3.schedule()}["S6 y: poisson rows"] == {t: b for t, b, _, _, _ in nav.schedule()}["S6 y: poisson rows"]
    nav3.update(3)
    f3 = nav3.physical_fields()
    for k in f1:
        assert K.rel(f1[k], f3[k]) < 1e-13, (k, K.rel(f1[k], f3[k]))


def test_s9_pressure_update_as_one_kernel(emu_lib, monkeypatch):
    """S9 of the confined step -- pres += to_ortho_x(S_y pseu) / dt - nu div and d/dx pres for the next step -- as the
    whole-line kernel csrc/pres_line.h: against the oracle (257-point x-lines with few and with many rows, 1025: one wave per
    line) and against the line program of the stage (RPDE_S9_LINE=0: the A/B switch), incl. the pressure and d/dx p."""
    K.check_step_parity(emu_lib, False, 257, 17, 1e5, 0.01, 3, check_at=[1, 3])
    K.check_step_parity(emu_lib, False, 257, 33, 1e5, 0.01, 2, check_at=[2])
    K.check_step_parity(emu_lib, False, 1025, 17, 1e6, 2e-3, 2, check_at=[2], eig_mode="shared")
    nav, _ = K.make_pair(emu_lib, False, 257, 17, 1e5, 1.0, 0.01, 1.0)
    assert {t: kind for t, _, _, _, kind in nav.schedule()}["S9 x: pressure update"] == "whole-line pressure update"
    nav.update(3)
    monkeypatch.setenv("RPDE_S9_LINE", "0")
    nav0, _ = K.make_pair(emu_lib, False, 257, 17, 1e5, 1.0, 0.01, 1.0)
    assert _has_line_program(nav0, "S9 x")
    nav0.update(3)
    for k in ("velx", "vely", "temp", "pres"):
        assert K.rel(getattr(nav, k).vhat, getattr(nav0, k).vhat) < 1e-12, (k, K.rel(getattr(nav, k).vhat, getattr(nav0, k).vhat))


def check_lines_of_2049_points(lib, monkeypatch, periodic, nx, steps=2):
    """y-lines of 2049 points (BASELINE config 5's, round 6): the velocity transforms of S2 and the three convection terms on the
    half-length core with two waves per line (hdct_line.h, N = 2048: passes 8 x 8 x 8 x 2), S6 as the whole-line row solve -- against
    the oracle and against the line programs of the same stages (RPDE_WHOLE_LINE=0)."""
    K.check_step_parity(lib, periodic, nx, 2049, 1e6, 1e-3, steps, check_at=[steps])
    nav, _ = K.make_pair(lib, periodic, nx, 2049, 1e6, 1.0, 1e-3, 1.0)
    kinds = {t: kind for t, _, _, _, kind in nav.schedule()}
    assert kinds["S2 y: velx -> phys"] == "whole-line transform" and kinds["S2 y: conv_temp"] == "whole-line convection term"
    assert kinds["S6 y: poisson rows"] == "whole-line poisson rows"
    nav.update(steps)
    monkeypatch.setenv("RPDE_WHOLE_LINE", "0")
    nav0, _ = K.make_pair(lib, periodic, nx, 2049, 1e6, 1.0, 1e-3, 1.0)
    assert not any(kind.startswith("whole-line") for _, _, _, _, kind in nav0.schedule())
    nav0.update(steps)
    monkeypatch.delenv("RPDE_WHOLE_LINE")
    for k in ("velx", "vely", "temp", "pres"):
        assert K.rel(getattr(nav, k).vhat, getattr(nav0, k).vhat) < 1e-10, (k, K.rel(getattr(nav, k).vhat, getattr(nav0, k).vhat))


@pytest.mark.parametrize("periodic,nx", [(True, 16), (False, 17)])
def test_step_with_lines_of_2049_points(emu_lib, monkeypatch, periodic, nx):
    check_lines_of_2049_points(emu_lib, monkeypatch, periodic, nx)


def check_periodic_rows_ab(lib, monkeypatch, nx, ny, steps=4, bc="rbc"):
    """S5 / S8 / S9 of the PERIODIC step as element-wise kernels (csrc/per_rows.h: one thread per complex number, the default)
    against the same stages as line programs (RPDE_PER_ROWS=0), same engine, same setup data; both meet the oracle."""
    K.check_step_parity(lib, True, nx, ny, 1e5, 0.01, steps, check_at=[1, steps], bc=bc)
    nav, _ = K.make_pair(lib, True, nx, ny, 1e5, 1.0, 0.01, 1.0, bc=bc)
    kinds = {t: kind for t, _, _, _, kind in nav.schedule()}
    assert all(kinds[t] == "element-wise rows" for t in ("S5 x: div", "S8 x: correction-x", "S9 x: pressure update")), kinds
    nav.update(steps)
    monkeypatch.setenv("RPDE_PER_ROWS", "0")
    nav0, _ = K.make_pair(lib, True, nx, ny, 1e5, 1.0, 0.01, 1.0, bc=bc)
    assert all(_has_line_program(nav0, t) for t in ("S5 x", "S8 x", "S9 x"))
    nav0.update(steps)
    monkeypatch.delenv("RPDE_PER_ROWS")
    assert nav.exit() is False and nav0.exit() is False
    for k in ("velx", "vely", "temp", "pres"):
        assert K.rel(getattr(nav, k).vhat, getattr(nav0, k).vhat) < 1e-12, (k, K.rel(getattr(nav, k).vhat, getattr(nav0, k).vhat))


@pytest.mark.parametrize("nx,ny,bc", [(64, 33, "rbc"), (18, 13, "rbc"), (256, 65, "rbc"), (32, 33, "hc")])
def test_periodic_elementwise_stages_equal_line_programs(emu_lib, monkeypatch, nx, ny, bc):
    check_periodic_rows_ab(emu_lib, monkeypatch, nx, ny, bc=bc)


def test_periodic_elementwise_stages_raise_the_nan_flag(emu_lib):
    """The stores of S8 / S9 are the device side of Integrate::exit (navier.rs:482-489): a NaN in the state must raise the flag."""
    nav, _ = K.make_pair(emu_lib, True, 32, 17, 1e5, 1.0, 0.01, 1.0)
    v = nav.velx.vhat.copy()
    v[3, 2] = np.nan
    nav.velx.vhat = v
    nav.update(1)
    assert nav.exit() is True


@pytest.mark.parametrize("periodic", [False, True])
def test_step_with_the_convection_terms_through_the_whole_line_kernel(emu_lib, monkeypatch, periodic):
    """conv_velx / conv_vely / conv_temp as three transforms per y-line in registers (csrc/dct_line.h conv_line);
    ny = 257 is a length the emulation build covers; RPDE_CONV_LINE=0 and RPDE_WHOLE_LINE=0 keep the line programs."""
    nx = 16 if periodic else 17
    K.check_step_parity(emu_lib, periodic, nx, 257, 1e5, 0.01, 4, check_at=[1, 4])
    nav, _ = K.make_pair(emu_lib, periodic, nx, 257, 1e5, 1.0, 0.01, 1.0)
    assert not _has_line_program(nav, "conv_temp") and not _has_line_program(nav, "S2 y: velx")
    monkeypatch.setenv("RPDE_CONV_LINE", "0")
    K.check_step_parity(emu_lib, periodic, nx, 257, 1e5, 0.01, 2)
    nav, _ = K.make_pair(emu_lib, periodic, nx, 257, 1e5, 1.0, 0.01, 1.0)
    assert _has_line_program(nav, "conv_velx") and not _has_line_program(nav, "S2 y: velx")
    monkeypatch.delenv("RPDE_CONV_LINE")
    monkeypatch.setenv("RPDE_WHOLE_LINE", "0")
    K.check_step_parity(emu_lib, periodic, nx, 257, 1e5, 0.01, 2)
    nav, _ = K.make_pair(emu_lib, periodic, nx, 257, 1e5, 1.0, 0.01, 1.0)
    assert _has_line_program(nav, "conv_velx") and _has_line_program(nav, "S2 y: velx")


@pytest.mark.parametrize("n,lift", [(257, False), (257, True), (4097, True)])
def test_conv_line_operator(emu_lib, n, lift):
    K.check_conv_line(emu_lib, n, nlines=3, lift=lift)


def test_step_with_every_whole_line_path(emu_lib):
    """257 x 257 confined: S1, the pure transforms of S2 and the convection terms all run csrc/dct_line.h."""
    K.check_step_parity(emu_lib, False, 257, 257, 1e6, 2e-3, 3, check_at=[1, 3])


def test_step_with_lines_of_1025_points(emu_lib):
    """N = 1024: one wave per line on the half-length core (hdct_line.h): S1, S3 (nx = 1025) and the pure transforms of
    S2 and the convection terms (ny = 1025: hconv_line, the three transforms of a term on the half-length core)."""
    # nx = 1025: both sides on the engine's eigen-decomposition, like at 4097 (two LAPACK runs
    # differ by 1e-9 in u after one step at this size -- the reference's own sensitivity)
    K.check_step_parity(emu_lib, False, 1025, 17, 1e6, 2e-3, 3, check_at=[1, 3], eig_mode="shared")
    nav, _ = K.make_pair(emu_lib, False, 1025, 17, 1e6, 1.0, 2e-3, 1.0)
    assert not _has_line_program(nav, "S1 x") and not _has_line_program(nav, "S3 x")
    K.check_step_parity(emu_lib, False, 17, 1025, 1e6, 2e-3, 2, check_at=[2])
    nav, _ = K.make_pair(emu_lib, False, 17, 1025, 1e6, 1.0, 2e-3, 1.0)
    assert not _has_line_program(nav, "S2 y: velx") and not _has_line_program(nav, "conv_velx") and not _has_line_program(nav, "conv_temp")


def test_periodic_step_with_lines_of_1025_points(emu_lib):
    """BASELINE configs[2] geometry in the y direction (periodic, ny = 1025): the pure transforms and the convection terms of
    S2 on the half-length core, batched over the fields."""
    K.check_step_parity(emu_lib, True, 16, 1025, 1e6, 2e-3, 3, check_at=[1, 3])
    nav, _ = K.make_pair(emu_lib, True, 16, 1025, 1e6, 1.0, 2e-3, 1.0)
    assert not _has_line_program(nav, "S2 y: velx") and not _has_line_program(nav, "conv_temp")


def test_whole_line_launches_of_short_lines_go_out_batched(emu_lib):
    """Lines the batched form covers (1025 points on the GPU, also 257 in the emulation build): the whole-line launches of
    the three fields of a stage are ONE launch (LineBatch, blockIdx.y = field) and the schedule says so: 17 launches
    instead of 24.  RPDE_LINE_BATCH (a bit mask per kind, read once per process) is the A/B switch; the results of the
    batched step are checked against the oracle by test_step_with_every_whole_line_path and the 257 / 1025 cases above."""
    nav, _ = K.make_pair(emu_lib, False, 257, 257, 1e6, 1.0, 2e-3, 1.0)
    sched = {t: kind for t, _, _, _, kind in nav.schedule()}
    assert sched["S1 x: state -> phys-x + d/dx"] == "whole-line transform pair (3 arrays)"
    assert sched["S2 y: velx -> phys + vely -> phys"] == "whole-line transform (2 arrays)"
    assert sched["S2 y: conv_velx + conv_vely + conv_temp"] == "whole-line convection term (3 arrays)"
    assert sched["S3 x: rhs + hholtz-x velx + vely + temp"] == "whole-line rhs + hholtz-x (3 arrays)"
    assert len(sched) == 16   # (round 6: pseu[0,0] = 0 rides in the store of G2)
    # the per-launch profile uses the same grouping and tags
    tags = [r["tag"] for r in nav.profile(1)]
    assert "S2 y: conv_velx + conv_vely + conv_temp" in tags and len(tags) == 16


def test_whole_line_launches_of_4097_point_lines_go_out_batched(emu_lib):
    """Round 5: lines of 4097 points take the batched form too (bits 4 - 7 of RPDE_LINE_BATCH; kernels.cc launch_line_batch_n<4096>:
    the full-length convection kernel and the three right-hand sides of S3 behind blockIdx.y) -- the drain of one field's last
    workgroups overlaps the start of the next field's.  9 x 4097: the pure transforms and the convection terms of S2;
    4097 x 9: S1 and S3; both against the oracle through the batched launches."""
    K.check_step_parity(emu_lib, False, 9, 4097, 1e6, 2e-3, 2, check_at=[2])
    nav, _ = K.make_pair(emu_lib, False, 9, 4097, 1e6, 1.0, 2e-3, 1.0)
    sched = {t: kind for t, _, _, _, kind in nav.schedule()}
    assert sched["S2 y: velx -> phys + vely -> phys"] == "whole-line transform (2 arrays)"
    assert sched["S2 y: conv_velx + conv_vely + conv_temp"] == "whole-line convection term (3 arrays)"
    K.check_step_parity(emu_lib, False, 4097, 9, 1e6, 2e-3, 2, check_at=[2], eig_mode="shared")
    nav, _ = K.make_pair(emu_lib, False, 4097, 9, 1e6, 1.0, 2e-3, 1.0)
    sched = {t: kind for t, _, _, _, kind in nav.schedule()}
    assert sched["S1 x: state -> phys-x + d/dx"] == "whole-line transform pair (3 arrays)"
    assert sched["S3 x: rhs + hholtz-x velx + vely + temp"] == "whole-line rhs + hholtz-x (3 arrays)"


def test_eig_cache(emu_lib, tmp_path, monkeypatch):
    """RPDE_EIG_CACHE: the second engine of an operator reads the decomposition the first one wrote (one file per pencil,
    named after its bytes) and steps bit-identically; a truncated file is ignored and replaced."""
    import glob
    monkeypatch.setenv("RPDE_EIG_CACHE", str(tmp_path))
    runs = []
    for attempt in range(3):
        nav = R.Navier2D.new_confined(65, 33, 1e5, 1.0, 0.01, 1.0, "rbc", library=emu_lib, init_random=None)
        nav.set_velocity(0.2, 1.0, 1.0); nav.set_temperature(0.2, 1.0, 1.0)
        nav.update(3)
        runs.append(nav.physical_fields())
        del nav
        files = glob.glob(str(tmp_path / "eigx_63_*.bin"))
        assert len(files) == 1, files
        if attempt == 1:                      # damage the file: the next engine falls back to LAPACK and rewrites it
            size = os.path.getsize(files[0])
            with open(files[0], "r+b") as f:
                f.truncate(size // 2)
        elif attempt == 2:
            assert os.path.getsize(files[0]) == size
    for k in runs[0]:
        assert np.array_equal(runs[0][k], runs[1][k]) and np.array_equal(runs[0][k], runs[2][k]), k


def test_confined_step_aspect(emu_lib):
    K.check_step_parity(emu_lib, False, 33, 17, 1e5, 0.01, 5, aspect=2.0)


@pytest.mark.parametrize("nx,ny,steps,aspect", [(16, 17, 5, 1.0), (32, 33, 10, 1.0), (64, 33, 10, 1.0), (32, 17, 5, 2.0),
                                                (16, 257, 3, 1.0)])   # ny = 257: whole-line kernel in S2
def test_periodic_step(emu_lib, nx, ny, steps, aspect):
    K.check_step_parity(emu_lib, True, nx, ny, 1e5, 0.01, steps, aspect=aspect, check_at=[1, 2, steps])


@pytest.mark.parametrize("nx", [8192, 16384])
def test_periodic_step_long_fourier_lines(emu_lib, monkeypatch, nx):
    """nx = 16384 (BASELINE config 5's line length) and 8192: the one-slot 1024-thread configuration
    with a 4096- / 8192-point complex FFT per x-line; aspect 8 as in that config."""
    K.check_step_parity(emu_lib, True, nx, 9, 1e5, 0.01, 2, aspect=8.0, check_at=[1, 2])
    # round 4: these lines run through the whole-line Fourier kernels as well (rfft_line.h: 8 x 8 x 8 x 8 (x 2) passes, the two
    # transforms of S1 one after the other in one buffer); A/B against the line programs
    nav, _ = K.make_pair(emu_lib, True, nx, 9, 1e5, 1.0, 0.01, 8.0)
    kinds = {t: kind for t, _, _, _, kind in nav.schedule()}
    assert kinds["S1 x: state -> phys-x + d/dx"].startswith("whole-line transform pair") and kinds["S3 x: rhs + hholtz-x vely"].startswith("whole-line rhs"), kinds
    nav.update(2)
    monkeypatch.setenv("RPDE_S1_LINE", "0")
    monkeypatch.setenv("RPDE_S3_LINE", "0")
    ref, _ = K.make_pair(emu_lib, True, nx, 9, 1e5, 1.0, 0.01, 8.0)
    assert _has_line_program(ref, "S1 x") and _has_line_program(ref, "S3 x")
    ref.update(2)
    for k in ("velx", "vely", "temp", "pres"):
        assert K.rel(getattr(nav, k).vhat, getattr(ref, k).vhat) < (1e-10 if k == "pres" else 1e-11), k


def test_errors_mirror_reference_panics(emu_lib):
    with pytest.raises(R.RpdeError, match="not recognized"):
        R.Navier2D.new_confined(17, 17, 1e4, 1.0, 0.01, 1.0, "xyz", library=emu_lib)
    nav = R.Navier2D.new_confined(17, 17, 1e4, 1.0, 0.01, 1.0, "rbc", library=emu_lib)
    with pytest.raises(R.RpdeError):
        nav.velx.v = np.zeros((3, 3))
    sp = R.Space2(R.cheb_dirichlet(9), R.cheb_dirichlet(9), library=emu_lib)
    with pytest.raises(R.RpdeError, match="length"):
        sp.forward(np.zeros((4, 4)))


def test_field_roundtrip_and_random_ic(emu_lib):
    nav = R.Navier2D.new_confined(17, 33, 1e4, 1.0, 0.01, 1.0, "rbc", library=emu_lib)
    rng = np.random.default_rng(3)
    vh = rng.standard_normal((15, 31))
    nav.temp.vhat = vh
    assert K.rel(nav.temp.vhat, vh) < 1e-15
    p = rng.standard_normal((17, 33))
    nav.pres.vhat = p
    assert K.rel(nav.pres.vhat, p) < 1e-15
    nav.init_random(0.1, seed=7)
    assert np.abs(nav.velx.v).max() < 0.2


def test_integrate_matches_reference_loop(emu_lib):
    """rustpde::integrate (src/lib.rs:187-219): stops once time + 1e-4 dt >= max_time."""
    nav, ora = K.make_pair(emu_lib, False, 17, 17, 1e4, 1.0, 0.01, 1.0)
    n_gpu = nav.integrate(0.075)
    n_ref = ora.integrate(0.075)
    assert n_gpu == n_ref == 8
    assert abs(nav.get_time() - ora.time) < 1e-12
    assert K.rel(nav.temp.vhat, ora.temp.vhat) < 1e-10
    # the python mirror of the loop drives update()/exit() one step at a time
    nav2, _ = K.make_pair(emu_lib, False, 17, 17, 1e4, 1.0, 0.01, 1.0)
    assert R.integrate(nav2, 0.075) == 8
    assert K.rel(nav2.temp.vhat, nav.temp.vhat) < 1e-14


def test_random_initial_condition_parity(emu_lib):
    """init_random of the reference is unseeded; parity runs inject the same arrays on both sides."""
    nav, ora = K.make_pair(emu_lib, False, 33, 17, 1e5, 1.0, 0.005, 1.0)
    rng = np.random.default_rng(11)
    for name in ("temp", "velx", "vely"):
        v = rng.uniform(-0.1, 0.1, size=(33, 17))
        getattr(nav, name).v = v
        ora.set_field_physical(name, v)
    for _ in range(10):
        nav.update(); ora.update()
    got, want = nav.physical_fields(), ora.physical_fields()
    for k in want:
        assert K.rel(got[k], want[k]) < 1e-10, k


@pytest.mark.parametrize("periodic,nx,ny,pr", [(False, 33, 17, 0.7), (True, 32, 17, 7.0)])
def test_prandtl_number_not_one(emu_lib, periodic, nx, ny, pr):
    """nu != ka: a swap of the two diffusivities cannot hide (every other case runs Pr = 1)."""
    K.check_step_parity(emu_lib, periodic, nx, ny, 1e5, 0.01, 6, pr=pr, check_at=[1, 6])


@pytest.mark.parametrize("periodic", [False, True])
def test_exit_flag(emu_lib, periodic):
    """Integrate::exit through the flag raised by the guarded stores of the step."""
    ctor = R.Navier2D.new_periodic if periodic else R.Navier2D.new_confined
    nav = ctor(32 if periodic else 33, 17, 1e5, 1.0, 0.01, 1.0, "rbc", library=emu_lib)
    nav.set_velocity(0.2, 1.0, 1.0)
    nav.set_temperature(0.2, 1.0, 1.0)
    assert nav.exit() is False
    nav.update(3)
    assert nav.exit() is False
    t = nav.temp.v
    t[5, 5] = np.nan
    nav.temp.v = t
    nav.update(1)
    assert nav.exit() is True and np.isnan(nav.div_norm())


def test_full_eigenbasis_solver(emu_lib):
    """Engine (parity-block eigenbasis) vs the oracle with the reference's single full dgeev."""
    K.check_solvers(emu_lib, "cheb_neumann", 33, "cheb_neumann", 17, [1.0, 1.0], eig_mode="full", poisson_tol=1e-6)


@pytest.mark.parametrize("k0,n0,k1,n1", [("cheb_dirichlet", 4097, "cheb_dirichlet", 9), ("cheb_neumann", 9, "cheb_dirichlet", 2049),
                                         ("cheb_dirichlet", 1025, "cheb_neumann", 17)])
def test_hholtz_long_lines(emu_lib, k0, n0, k1, n1):
    """The wave-serial scans of the Helmholtz Fdma solve at the line lengths of the BASELINE configs
    (chunk lengths 33, 17, 9 per lane)."""
    sp, osp = K.spaces(emu_lib, k0, n0, k1, n1)
    rhs = np.random.default_rng(2).standard_normal(osp.shape_ortho)
    for c in ([2e-8, 2e-8], [1e-3, 5e-4]):
        assert K.rel(R.HholtzAdi(sp, c).solve(rhs), K.S.HholtzAdi(osp, c).solve(rhs)) < 1e-11


def test_poisson_513(emu_lib):
    K.check_solvers(emu_lib, "cheb_neumann", 513, "cheb_neumann", 9, [1.0, 1.0])


@pytest.mark.parametrize("periodic,nx,ny", [(False, 17, 129), (False, 9, 257), (True, 16, 257)])
def test_column_scans_several_blocks(emu_lib, periodic, nx, ny):
    """The Helmholtz-y solve and the y-derivatives as column scans (colscan.h) with more than one block of
    64 rows: block carries, partial last block (127 = 64 + 63 rows, 255 = 3 x 64 + 63)."""
    K.check_step_parity(emu_lib, periodic, nx, ny, 1e5, 0.01, 4, check_at=[1, 4])


@pytest.mark.parametrize("periodic,nx,ny,bc,structured", [(False, 257, 65, "rbc", True), (False, 65, 257, "rbc", True),
                                                            (True, 256, 65, "rbc", True), (False, 257, 65, "hc", False)])
def test_lift_structure_is_found_and_changes_no_bit(emu_lib, monkeypatch, periodic, nx, ny, bc, structured):
    """Round 6: the engine looks ONCE at the time-independent arrays of the boundary-condition lift (its physical gradients in the
    temperature's convection term, its spectral rows and its Laplacian in the right-hand sides of vely / temp) and reads only what
    is not redundant -- for "rbc" (a lift that does not depend on x) one y-line of the gradients for all lines and one leading
    coefficient per spectral row, nothing of the vanishing Laplacian; "hc" (a lift that varies along x) finds nothing to skip.
    RPDE_LIFT_STRUCT=0 reads whole arrays: the fields must agree bit for bit, and the schedule's byte count shows what was found."""
    ctor = R.Navier2D.new_periodic if periodic else R.Navier2D.new_confined
    out = {}
    for flag in ("1", "0"):
        if flag == "0":
            monkeypatch.setenv("RPDE_LIFT_STRUCT", "0")
        else:
            monkeypatch.delenv("RPDE_LIFT_STRUCT", raising=False)
        nav = ctor(nx, ny, 1e5, 1.0, 0.01, 1.0, bc, library=emu_lib, init_random=None)
        nav.set_velocity(0.2, 1.0, 1.0)
        nav.set_temperature(0.2, 1.0, 1.0)
        nav.update(4)
        out[flag] = (nav.physical_fields(), sum(row[1] for row in nav.schedule()))
    monkeypatch.delenv("RPDE_LIFT_STRUCT", raising=False)
    for k in out["1"][0]:
        assert np.array_equal(out["1"][0][k], out["0"][0][k]), k
    if structured:
        assert out["1"][1] < out["0"][1], (out["1"][1], out["0"][1])
    else:
        assert out["1"][1] == out["0"][1]
