"""The "hc" boundary condition (horizontal convection): the temperature lives in
cheb_neumann(nx) x cheb_dirichlet_neumann(ny) (navier.rs:245-248 / 366-369), its lift is bc_hc
(boundary_conditions.rs:96-134 / 163-202) and its Helmholtz solve along y is PdmaPlus2 (pdma_plus2.rs:45-157).

How the oracle is pinned here: cheb_dirichlet_neumann lives in the absent funspace crate and no test of the reference
holds a known answer for it, so the stencil is pinned by what it must do (every basis function vanishes at y = -1 and
has zero slope at y = +1 -- the walls the reference's lift bc_hc assigns), PdmaPlus2 by the reference's own test
(test_pdma_dim1: the matrix times the solution recovers the right-hand side) and against a dense solve, the band forms
against dense products, and the whole Helmholtz chain by an analytic solution that satisfies the mixed boundary
conditions (like hholtz_adi.rs:248-277 does for Dirichlet walls)."""
import numpy as np
import pytest

import rustpde_mpi_amd as R
from oracle import bases as B, navier as N, solver as S
from tests import checks as K


# ----------------------------------------------------------------------------------------------- oracle pins (CPU)
@pytest.mark.parametrize("n", [9, 33, 130])
def test_oracle_stencil_satisfies_the_mixed_boundary_conditions(n):
    b = B.cheb_dirichlet_neumann(n)
    a = np.random.default_rng(n).standard_normal((b.m, 4))
    c = b.to_ortho(a, 0)
    o = B.chebyshev(n)
    v = o.backward_ortho(c, 0)
    dv = o.backward_ortho(o._cheb_diff0(c, 1), 0)
    scale = np.abs(v).max()
    assert np.abs(v[0]).max() < 1e-12 * scale          # Dirichlet at y[0] = -1 (the bottom wall of bc_hc)
    assert np.abs(dv[-1]).max() < 1e-9 * np.abs(dv).max()   # Neumann at y[n-1] = +1
    assert np.abs(v[-1]).max() > 1e-3 * scale          # ... and it is not Dirichlet there
    # from_ortho is the least-squares projection: exact on the span of the stencil, S^T residual = 0 off it
    assert K.rel(b.from_ortho(c, 0), a) < 1e-12
    r = np.random.default_rng(1).standard_normal((n, 3))
    proj = b.to_ortho(b.from_ortho(r, 0), 0)
    assert np.abs(b.mass_dense().T @ (r - proj)).max() < 1e-11


def test_oracle_pdma_plus2_reference_test():
    """test_pdma_dim1 (pdma_plus2.rs:215-251): the matrix times the solution recovers the data (1e-3 there)."""
    nx = 6
    m = np.zeros((nx, nx))
    for i in range(nx):
        j = i + 1.0
        m[i, i] = 0.5 * j
        if i > 1: m[i, i - 2] = 10.0 * j
        if i > 0: m[i, i - 1] = 4.0 * j
        if i < nx - 1: m[i, i + 1] = 1.5 * j
        if i < nx - 2: m[i, i + 2] = 3.5 * j
        if i < nx - 3: m[i, i + 3] = 4.5 * j
        if i < nx - 4: m[i, i + 4] = 2.5 * j
    bands = {o: np.array([m[r, r + o] if 0 <= r + o < nx else 0.0 for r in range(nx)]) for o in range(-2, 5)}
    data = np.arange(nx, dtype=float)
    x = S.PdmaPlus2(bands).solve(data[:, None], 0)[:, 0]
    assert np.abs(m @ x - data).max() < 1e-12
    assert np.abs(x - np.linalg.solve(m, data)).max() < 1e-13


@pytest.mark.parametrize("n", [8, 21, 64])
def test_oracle_pdma_plus2_random_bands(n):
    rng = np.random.default_rng(n)
    m = np.zeros((n, n))
    for o in range(-2, 5):
        for r in range(n):
            if 0 <= r + o < n:
                m[r, r + o] = rng.standard_normal() + (6.0 if o == 0 else 0.0)
    bands = {o: np.array([m[r, r + o] if 0 <= r + o < n else 0.0 for r in range(n)]) for o in range(-2, 5)}
    rhs = rng.standard_normal((n, 3))
    assert K.rel(S.PdmaPlus2(bands).solve(rhs, 0), np.linalg.solve(m, rhs)) < 1e-12


def test_oracle_seven_band_forms_match_dense_products():
    b = B.cheb_dirichlet_neumann(19)
    ma, mb = b.hholtz_bands7()
    peye, b2, s = b.laplace_inv_eye_dense(), b.laplace_inv_dense(), b.mass_dense()

    def dense(bands):
        a = np.zeros((b.m, b.m))
        for o, arr in bands.items():
            for r in range(b.m):
                if 0 <= r + o < b.m:
                    a[r, r + o] = arr[r]
        return a
    assert np.abs(dense(ma) - peye @ b2 @ s).max() < 1e-15
    assert np.abs(dense(mb) - peye @ s).max() < 1e-15
    # a two-term stencil through the same routine reproduces the four-diagonal forms
    d = B.cheb_dirichlet(19)
    ma7, mb7 = d.hholtz_bands7()
    (lo, di, u1, u2), (bd, bu) = d.hholtz_bands()
    for got, want in ((ma7[-2], lo), (ma7[0], di), (ma7[2], u1), (ma7[4], u2), (mb7[0], bd), (mb7[2], bu)):
        assert np.abs(got - want).max() < 1e-16
    assert all(np.abs(ma7[o]).max() == 0.0 for o in (-1, 1, 3))


def test_oracle_helmholtz_analytic_mixed_walls():
    """(1 - c d2/dy2) u = f with u = sin(pi (y + 1) / 4): u(-1) = 0, u'(1) = 0."""
    n, c = 65, 0.3
    b = B.cheb_dirichlet_neumann(n)
    y = b.coords()
    u = np.sin(np.pi * (y + 1.0) / 4.0)
    f = u * (1.0 + c * np.pi ** 2 / 16.0)
    fh = B.chebyshev(n).forward_ortho(f[:, None], 0)[:, 0]
    x = S.HholtzAdi1(b, c).solve(fh)
    assert np.abs(b.backward(x[:, None], 0)[:, 0] - u).max() < 1e-13


@pytest.mark.parametrize("periodic", [False, True])
def test_oracle_hc_run_is_physical(periodic):
    """A short run: the total temperature keeps the bottom profile of the lift and an insulated top, the flow stays
    divergence free to the splitting error, nothing blows up."""
    ctor = N.Navier2D.new_periodic if periodic else N.Navier2D.new_confined
    nav = ctor(32 if periodic else 33, 33, 1e5, 1.0, 0.01, 1.0, "hc", eig_mode="parity")
    nav.set_velocity(0.2, 1.0, 1.0)
    nav.set_temperature(0.2, 1.0, 1.0)
    for _ in range(100):
        nav.update()
    f = nav.physical_fields()
    total = f["temp"] + nav.tempbc.v
    x = nav.tempbc.x[0]
    assert np.abs(total[:, 0] + 0.5 * np.cos(2.0 * np.pi * (x - x[0]) / (x[-1] - x[0]))).max() < 1e-12
    fld = nav.field
    fld.vhat = (nav.temp.to_ortho() + nav.tempbc.to_ortho())
    dtdy = fld.space.backward(fld.gradient([0, 1], None))
    assert np.abs(dtdy[:, -1]).max() < 1e-8 * max(1.0, np.abs(dtdy).max())
    assert all(np.isfinite(v).all() and np.abs(v).max() < 2.0 for v in f.values())
    assert nav.div_norm() < 1e-2
    with pytest.raises(ValueError, match="not recognized"):
        ctor(16, 17, 1e4, 1.0, 0.01, 1.0, "xyz")


# ----------------------------------------------------------------------------------------------- kernel sources (CPU emulation)
HC_SPACES = [("cheb_neumann", 17, "cheb_dirichlet_neumann", 33), ("fourier_r2c", 16, "cheb_dirichlet_neumann", 9),
             ("cheb_neumann", 9, "cheb_dirichlet_neumann", 300), ("cheb_neumann", 9, "cheb_dirichlet_neumann", 1025),
             ("fourier_r2c", 64, "cheb_dirichlet_neumann", 65)]
HC_SOLVERS = [("cheb_neumann", 33, "cheb_dirichlet_neumann", 17, [1e-3, 2e-3]),
              ("cheb_neumann", 65, "cheb_dirichlet_neumann", 129, [3e-5, 1e-5]),
              ("fourier_r2c", 64, "cheb_dirichlet_neumann", 33, [1e-3, 1.0])]
HC_STEPS = [(False, 17, 17, 5, 1.0), (False, 33, 33, 20, 1.0), (False, 65, 33, 10, 1.0), (False, 33, 65, 10, 2.0),
            (False, 17, 257, 4, 1.0), (False, 257, 17, 4, 1.0),     # 257: the velocities run the whole-line kernels
            (True, 16, 17, 5, 1.0), (True, 32, 33, 10, 1.0), (True, 64, 33, 10, 2.0), (True, 16, 257, 3, 1.0)]


def hholtz_only(lib, k0, n0, k1, n1, c, tol=1e-11):
    sp, osp = K.spaces(lib, k0, n0, k1, n1)
    rng = np.random.default_rng(5)
    rhs = rng.standard_normal(osp.shape_ortho)
    if k0 == "fourier_r2c":
        rhs = rhs + 1j * rng.standard_normal(osp.shape_ortho)
    e = K.rel(R.HholtzAdi(sp, c).solve(rhs), S.HholtzAdi(osp, c).solve(rhs))
    assert e < tol, e


@pytest.mark.parametrize("k0,n0,k1,n1", HC_SPACES)
def test_space_ops_three_term_axis(emu_lib, k0, n0, k1, n1):
    K.check_space_ops(emu_lib, k0, n0, k1, n1)


@pytest.mark.parametrize("k0,n0,k1,n1,c", HC_SOLVERS)
def test_hholtz_three_term_axis(emu_lib, k0, n0, k1, n1, c):
    hholtz_only(emu_lib, k0, n0, k1, n1, c)


@pytest.mark.parametrize("periodic,nx,ny,steps,aspect", HC_STEPS)
def test_hc_step(emu_lib, periodic, nx, ny, steps, aspect):
    K.check_step_parity(emu_lib, periodic, nx, ny, 1e5, 0.01, steps, aspect=aspect, check_at=[1, 2, steps], bc="hc")


def hc_blocked_ab(lib, periodic, nx, ny, ra, dt, steps, monkeypatch, tol=1e-12):
    """The blocked PdmaPlus2 column solve (pdma.h: blocks of 32 rows from zero inflow + tabulated homogeneous solutions) against
    one thread per column over all rows (RPDE_HC_BLOCKED=0): the same step to round-off."""
    runs = []
    for sw in ("1", "0"):
        monkeypatch.setenv("RPDE_HC_BLOCKED", sw)
        ctor = R.Navier2D.new_periodic if periodic else R.Navier2D.new_confined
        nav = ctor(nx, ny, ra, 1.0, dt, 1.0, "hc", library=lib)
        nav.set_velocity(0.2, 1.0, 1.0)
        nav.set_temperature(0.2, 1.0, 1.0)
        nav.update(steps)
        runs.append({k: np.array(getattr(nav, k).vhat) for k in ("velx", "vely", "temp", "pres")})
        assert nav.exit() is False
    for k in runs[0]:
        e = K.rel(runs[0][k], runs[1][k])
        assert e < tol, (k, e)
    return runs[0]


# my = ny - 2 rows: 33 = 32 + 1, 34 = 32 + 2, 35 = 32 + 3 (short last blocks: the backward inflows shine through the block),
# 64 (exact), 97 = 3 x 32 + 1, 1023 = 31 x 32 + 31
@pytest.mark.parametrize("periodic,nx,ny", [(False, 17, 35), (False, 17, 36), (False, 17, 37), (True, 16, 66), (False, 9, 99),
                                            (False, 17, 1025), (False, 9, 5), (False, 9, 4097)])
def test_hc_blocked_column_solve_equals_the_serial_one(emu_lib, monkeypatch, periodic, nx, ny):
    hc_blocked_ab(emu_lib, periodic, nx, ny, 1e5, 0.01, 3, monkeypatch)


def test_hc_schedule_and_restrictions(emu_lib):
    nav = R.Navier2D.new_confined(33, 33, 1e5, 1.0, 0.01, 1.0, "hc", library=emu_lib)
    kinds = {t: kind for t, _, _, _, kind in nav.schedule()}
    assert any(k == "row stencil" for k in kinds.values()) and any(k == "column solve" for k in kinds.values())
    # the space rejects what it cannot do instead of computing something else
    with pytest.raises(R.RpdeError, match="unknown base kind"):
        R.Space2(R.cheb_dirichlet_neumann(9), R.cheb_dirichlet(9), library=emu_lib)
    sp = R.Space2(R.cheb_neumann(9), R.cheb_dirichlet_neumann(9), library=emu_lib)
    with pytest.raises(R.RpdeError, match="two-term"):
        R.Poisson(sp, [1.0, 1.0])


def test_hc_snapshot_restart_and_statistics(emu_lib, tmp_path):
    """write / read (navier_io.rs:44-62), the callback diagnostics and the Statistics hook with the hc temperature."""
    from oracle import navier as NN
    nav, ora = K.make_pair(emu_lib, False, 33, 17, 1e5, 1.0, 0.01, 1.0, bc="hc")
    nav.update(3)
    f = str(tmp_path / "flow.h5")
    nav.write(f)
    nav2 = R.Navier2D.new_confined(33, 17, 1e5, 1.0, 0.01, 1.0, "hc", library=emu_lib)
    nav2.read(f)
    for k in ("velx", "vely", "temp", "pres"):
        assert K.rel(getattr(nav2, k).vhat, getattr(nav, k).vhat) < 1e-15, k
    nav.update(2)
    nav2.update(2)
    assert K.rel(nav2.temp.vhat, nav.temp.vhat) < 1e-13
    for _ in range(5):
        ora.update()
    st, so = R.Statistics.new(nav, 0.01, 1.0), NN.Statistics(ora, 0.01, 1.0)
    st.update()
    so.update_from(ora)
    for got, want in ((st.t_avg, so.t_avg), (st.nusselt, so.nusselt)):
        assert K.rel(got.vhat, want.vhat) < 1e-11


# ----------------------------------------------------------------------------------------------- the HIP kernels
@pytest.mark.gpu
@pytest.mark.parametrize("k0,n0,k1,n1", HC_SPACES + [("cheb_neumann", 65, "cheb_dirichlet_neumann", 4097)])
def test_space_ops_three_term_axis_gpu(hip_lib, k0, n0, k1, n1):
    K.check_space_ops(hip_lib, k0, n0, k1, n1)


@pytest.mark.gpu
@pytest.mark.parametrize("k0,n0,k1,n1,c", HC_SOLVERS + [("cheb_neumann", 65, "cheb_dirichlet_neumann", 4097, [2e-8, 2e-8])])
def test_hholtz_three_term_axis_gpu(hip_lib, k0, n0, k1, n1, c):
    hholtz_only(hip_lib, k0, n0, k1, n1, c, tol=2e-10 if n1 == 4097 else 1e-11)


@pytest.mark.gpu
@pytest.mark.parametrize("periodic,nx,ny,steps,aspect", HC_STEPS + [(False, 129, 129, 100, 1.0), (True, 256, 129, 20, 1.0)])
def test_hc_step_gpu(hip_lib, periodic, nx, ny, steps, aspect):
    K.check_step_parity(hip_lib, periodic, nx, ny, 1e5, 0.01, steps, aspect=aspect, check_at=[1, 2, steps], bc="hc")


@pytest.mark.gpu
def test_hc_step_1025_gpu(hip_lib):
    """1025 x 1025 (one-wave whole-line kernels for the velocities, batched column solve over 65 blocks of rows for the
    temperature), Ra = 1e7, dt = 1e-3; shared eigen-decomposition like every large confined comparison."""
    K.run_isolated('check_step_parity(lib, False, 1025, 1025, 1e7, 1e-3, 3, check_at=[1, 3], bc="hc", eig_mode="shared")')


@pytest.mark.gpu
@pytest.mark.parametrize("periodic,nx,ny,ra,dt", [(False, 17, 35, 1e5, 0.01), (False, 65, 99, 1e5, 0.01), (True, 64, 66, 1e5, 0.01),
                                                  (False, 257, 1025, 1e7, 1e-3), (False, 129, 4097, 1e8, 2e-4)])
def test_hc_blocked_column_solve_equals_the_serial_one_gpu(hip_lib, monkeypatch, periodic, nx, ny, ra, dt):
    # 4097 rows at Ra = 1e8: the pressure carries the round-off of the two orders of summation at 1.1e-12 (u, v, T below 1e-12)
    hc_blocked_ab(hip_lib, periodic, nx, ny, ra, dt, 3, monkeypatch, tol=1e-11 if ny == 4097 else 1e-12)


@pytest.mark.gpu
def test_hc_exit_flag_and_io_gpu(hip_lib, tmp_path):
    nav = R.Navier2D.new_confined(65, 65, 1e5, 1.0, 0.01, 1.0, "hc", library=hip_lib)
    nav.set_velocity(0.2, 1.0, 1.0)
    nav.set_temperature(0.2, 1.0, 1.0)
    nav.update(2)
    assert nav.exit() is False
    f = str(tmp_path / "flow.h5")
    nav.write(f)
    nav2 = R.Navier2D.new_confined(65, 65, 1e5, 1.0, 0.01, 1.0, "hc", library=hip_lib)
    nav2.read(f)
    assert K.rel(nav2.temp.vhat, nav.temp.vhat) < 1e-15
    t = nav.temp.vhat
    t[3, 3] = np.nan
    nav.temp.vhat = t
    nav.update(1)
    assert nav.exit() is True
