"""Navier2DAdjoint (first slice): the adjoint-descent solver of src/navier_stokes/steady_adjoint.rs.

oracle/adjoint.py restates `update()` (steady_adjoint.rs:541-608) and its equations (steady_adjoint_eq.rs); the tensor
Helmholtz solver behind the residual norm (src/solver/hholtz.rs) is pinned by the reference's own analytic tests
(tests/test_oracle_golden.py::test_hholtz_tensor_analytic).  The reference holds no golden for the adjoint step; what pins
the oracle's step beyond its solvers is the property the method is built on: the descent reduces the residual norm
(Farazmand 2016), and a steady state (conduction, below the critical Rayleigh number) is a fixed point.

Engine (csrc/adjoint.cc, C ABI rpde_adjoint2d_*) against that oracle: emulation build on the CPU, HIP build on the GPU
(129^2, 257^2, 1025^2, periodic 256 x 129)."""
import os

import numpy as np
import pytest

import rustpde_mpi_amd as R
from oracle import adjoint as A, bases as B, navier as N, solver as S

FIELDS = ("velx", "vely", "temp", "pres", "velx_adj", "vely_adj", "temp_adj", "pres_adj", "pseu")


def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def check_adjoint_parity(lib, nx, ny, periodic, steps, ra=1e4, dt=0.005, tol=1e-10, tol_p=1e-8):
    """Engine vs oracle after every update: u, v, T and the three adjoint fields to `tol` (BASELINE's 1e-10), the pressures
    (pres, pres_adj, pseu: they carry the Poisson solve's amplified eigenvector round-off) to `tol_p`;
    div_norm, norm_residual and exit() agree."""
    mk_e = R.Navier2DAdjoint.new_periodic if periodic else R.Navier2DAdjoint.new_confined
    mk_o = A.Navier2DAdjoint.new_periodic if periodic else A.Navier2DAdjoint.new_confined
    nav = mk_e(nx, ny, ra, 1.0, dt, 1.0, "rbc", library=lib)
    ora = mk_o(nx, ny, ra, 1.0, dt, 1.0, "rbc", eig_mode="parity")
    for z in (nav, ora):
        z.set_velocity(0.1, 1.0, 1.0)
        z.set_temperature(0.1, 1.0, 1.0)
    worst = {}
    for s in range(steps):
        nav.update(1)
        ora.update()
        got, want = nav.spectral_fields(FIELDS), ora.spectral_fields(FIELDS)
        for k in FIELDS:
            e = rel(got[k], want[k])
            worst[k] = max(worst.get(k, 0.0), e)
            assert e < (tol_p if k in ("pres", "pres_adj", "pseu") else tol), (nx, ny, periodic, s, k, e)
    assert abs(nav.get_time() - ora.time) < 1e-12 and abs(nav.get_time() - steps * dt) < 1e-12
    ro, rn = ora.norm_residual(), nav.norm_residual()
    for a, b in zip(rn, ro):
        assert abs(a - b) < 1e-9 * max(b, 1e-30), (rn, ro)
    assert abs(nav.div_norm() - ora.div_norm()) < 1e-8 * max(ora.div_norm(), 1e-12)
    assert nav.exit() == ora.exit()
    phys_e, phys_o = nav.physical_fields(), ora.physical_fields()
    for k in ("velx", "vely", "temp"):
        assert rel(phys_e[k], phys_o[k]) < tol, k
    print(nx, ny, "periodic" if periodic else "confined", {k: f"{v:.1e}" for k, v in worst.items()})
    return worst


def check_hholtz_operator(lib, n0=33, n1=29):
    """rpde_hholtz_* against the oracle's Hholtz on random smooth data, and the reference's analytic test
    (hholtz.rs:226-255) through the device operator."""
    rng = np.random.default_rng(5)
    for base0, ob0 in (((R.CHEB_DIRICHLET, n0), B.cheb_dirichlet(n0)), ((R.CHEB_NEUMANN, n0), B.cheb_neumann(n0)),
                       ((R.FOURIER_R2C, n0 - 1), B.fourier_r2c(n0 - 1))):
        sp = R.Space2(base0, (R.CHEB_DIRICHLET, n1), library=lib)
        osp = B.Space2(ob0, B.cheb_dirichlet(n1))
        f = N.Field2(osp)
        f.v = rng.standard_normal(f.v.shape)
        f.forward()
        rhs = f.to_ortho()
        for c in ([1e-1, 1e-1], [2.5e-2, 1e-1]):
            want = S.Hholtz(osp, c, eig_mode="parity").solve(rhs)
            got = R.Hholtz(sp, c).solve(rhs)
            assert rel(got, want) < 1e-11, (base0, c, rel(got, want))
    # hholtz.rs:226-255 (64 x 64, alpha = 1): cos(pi x / 2) cos(pi y / 2) / (1 + 2 alpha (pi / 2)^2)
    n, alpha, k = 64, 1.0, np.pi / 2
    sp = R.Space2((R.CHEB_DIRICHLET, n), (R.CHEB_DIRICHLET, n), library=lib)
    x = N.Field2(B.Space2(B.cheb_dirichlet(n), B.cheb_dirichlet(n))).x[0]
    v = np.cos(k * x)[:, None] * np.cos(k * x)[None, :]
    out = sp.backward(R.Hholtz(sp, [alpha, alpha]).solve(sp.to_ortho(sp.forward(v))))
    assert np.abs(out - v / (1 + alpha * k * k * 2)).max() < 1e-12


# ------------------------------------------------------------------------------------------------ oracle
def test_oracle_descent_reduces_the_residual():
    """33 x 33, Ra = 1e4, dt = 5e-3 (inside the explicit scheme's stability bound 0.2 DT_NAVIER / nu): the residual norm of u,
    v and T falls monotonically over 40 updates -- adjoint descent does what it is built for."""
    nav = A.Navier2DAdjoint.new_confined(33, 33, 1e4, 1.0, 0.005, 1.0, "rbc", eig_mode="parity")
    nav.set_velocity(0.1, 1.0, 1.0)
    nav.set_temperature(0.1, 1.0, 1.0)
    hist = []
    for _ in range(40):
        nav.update()
        hist.append(sum(nav.norm_residual()))
    assert all(b < a for a, b in zip(hist[2:], hist[3:])), hist
    assert hist[-1] < 0.5 * hist[2]
    assert not nav.exit()


def test_oracle_conduction_state_is_a_fixed_point():
    """Zero perturbation = the conduction state: no residual, nothing moves, exit() reports convergence
    (steady_adjoint.rs:631-635)."""
    nav = A.Navier2DAdjoint.new_confined(17, 17, 1e3, 1.0, 0.01, 1.0, "rbc", eig_mode="parity")
    nav.update()
    assert max(nav.norm_residual()) < 1e-12
    for k in ("velx", "vely", "temp"):
        assert np.abs(getattr(nav, k).vhat).max() < 1e-12
    assert nav.exit()


def test_oracle_rejects_unknown_bc():
    with pytest.raises(ValueError):
        A.Navier2DAdjoint.new_confined(17, 17, 1e3, 1.0, 0.01, 1.0, "xx")


# ------------------------------------------------------------------------------------------------ emulation build (CPU)
@pytest.mark.parametrize("nx,ny,periodic", [(33, 33, False), (65, 33, False), (32, 33, True), (64, 17, True)])
def test_emu_adjoint_step_parity(emu_lib, nx, ny, periodic):
    check_adjoint_parity(emu_lib, nx, ny, periodic, steps=4)


def test_emu_hholtz_operator(emu_lib):
    check_hholtz_operator(emu_lib)


def test_emu_adjoint_errors_and_fields(emu_lib):
    with pytest.raises(R.RpdeError, match="not supported"):
        R.Navier2DAdjoint.new_confined(17, 17, 1e4, 1.0, 0.01, 1.0, "hc", library=emu_lib)
    with pytest.raises(R.RpdeError, match="not recognized"):
        R.Navier2DAdjoint.new_confined(17, 17, 1e4, 1.0, 0.01, 1.0, "xx", library=emu_lib)
    nav = R.Navier2DAdjoint.new_confined(17, 19, 1e4, 1.0, 0.01, 1.0, "rbc", library=emu_lib)
    ora = A.Navier2DAdjoint.new_confined(17, 19, 1e4, 1.0, 0.01, 1.0, "rbc", eig_mode="parity")
    assert rel(nav.tempbc.vhat, ora.tempbc.vhat) < 1e-13                       # the lift
    p = nav.params
    assert abs(p["nu"] - ora.params["nu"]) < 1e-15 and abs(p["ka"] - ora.params["ka"]) < 1e-15
    assert nav.get_dt() == 0.01 and nav.get_time() == 0.0
    with pytest.raises(R.RpdeError, match="tempbc is fixed"):
        nav.tempbc.vhat = ora.tempbc.vhat
    with pytest.raises(R.RpdeError, match="unknown field"):
        R._FieldView(nav, "nope").vhat
    # spectral and physical set / get round trip, conduction state is a fixed point, exit() = converged
    nav.update(1)
    assert max(nav.norm_residual()) < 1e-12 and nav.exit()
    rng = np.random.default_rng(0)
    v = rng.standard_normal((17, 19))
    nav.velx.v = v
    ora.set_field_physical("velx", v)
    assert rel(nav.velx.vhat, ora.velx.vhat) < 1e-12
    nav.reset_time()
    assert nav.get_time() == 0.0


def test_emu_integrate_drives_the_adjoint_solver(emu_lib):
    """rustpde::integrate(&mut navier_adjoint, max_time, None) (examples/navier_rbc_steady.rs): update / exit loop."""
    nav = R.Navier2DAdjoint.new_confined(17, 17, 1e4, 1.0, 0.005, 1.0, "rbc", library=emu_lib)
    nav.set_velocity(0.1, 1.0, 1.0)
    nav.set_temperature(0.1, 1.0, 1.0)
    R.integrate(nav, 0.05, None)
    assert abs(nav.get_time() - 0.05) < 1e-9
    assert np.isfinite(nav.div_norm())


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("nx,ny,periodic,steps", [(129, 129, False, 5), (257, 257, False, 4), (256, 129, True, 4), (1025, 1025, False, 2)])
def test_gpu_adjoint_step_parity(hip_lib, nx, ny, periodic, steps):
    """Ra = 1e6 at 1025^2 keeps dt = 0.002 inside 0.2 DT_NAVIER / nu."""
    if nx >= 1025:
        check_adjoint_parity(hip_lib, nx, ny, periodic, steps, ra=1e6, dt=0.002)
    else:
        check_adjoint_parity(hip_lib, nx, ny, periodic, steps)


@pytest.mark.gpu
def test_gpu_hholtz_operator(hip_lib):
    check_hholtz_operator(hip_lib)
    check_hholtz_operator(hip_lib, 129, 65)


@pytest.mark.gpu
def test_gpu_descent_reduces_the_residual(hip_lib):
    nav = R.Navier2DAdjoint.new_confined(65, 65, 1e4, 1.0, 0.002, 1.0, "rbc", library=hip_lib)
    nav.set_velocity(0.1, 1.0, 1.0)
    nav.set_temperature(0.1, 1.0, 1.0)
    hist = []
    for _ in range(30):
        nav.update(1)
        hist.append(sum(nav.norm_residual()))
    assert all(b < a for a, b in zip(hist[2:], hist[3:])), hist


# ------------------------------------------------------------------------------------------------ snapshots (steady_adjoint_io.rs)
@pytest.mark.parametrize("periodic", [False, True])
def test_emu_adjoint_write_read(emu_lib, tmp_path, periodic):
    """Navier2DAdjoint::write / read: the Field2 layout of Navier2D::write (independent parser tests/h5classic.py); read takes
    ux, uy, temp and time; a restart continues bit-identically in u, v, T (pres / pres_adj are not part of the file's
    restart set, steady_adjoint_io.rs:22-33)."""
    from tests.h5classic import File
    mk = R.Navier2DAdjoint.new_periodic if periodic else R.Navier2DAdjoint.new_confined
    nx = 16 if periodic else 17
    nav = mk(nx, 17, 1e4, 1.0, 0.005, 1.0, "rbc", library=emu_lib)
    nav.set_velocity(0.1, 1.0, 1.0)
    nav.set_temperature(0.1, 1.0, 1.0)
    nav.update(2)
    fn = str(tmp_path / "adjoint.h5")
    nav.write(fn)
    f = File(fn).datasets
    groups = ("ux", "uy", "temp", "pres", "tempbc")
    spec = ("vhat_re", "vhat_im") if periodic else ("vhat",)
    assert sorted(f) == sorted([f"{g}/{d}" for g in groups for d in ("x", "dx", "y", "dy", "v") + spec] + ["time", "ra", "pr", "nu", "ka"])
    for g, name in zip(groups, ("velx", "vely", "temp", "pres", "tempbc")):
        fld = getattr(nav, name)
        assert np.array_equal(f[g + "/v"], fld.v)
        if periodic:
            assert np.array_equal(f[g + "/vhat_re"], fld.vhat.real) and np.array_equal(f[g + "/vhat_im"], fld.vhat.imag)
        else:
            assert np.array_equal(f[g + "/vhat"], fld.vhat)
    assert f["time"][0] == nav.get_time() and f["ra"][0] == 1e4
    nav2 = mk(nx, 17, 1e4, 1.0, 0.005, 1.0, "rbc", library=emu_lib)
    nav2.read(fn)
    assert nav2.get_time() == nav.get_time()
    for k in ("velx", "vely", "temp"):
        assert np.array_equal(getattr(nav, k).vhat, getattr(nav2, k).vhat), k
    with pytest.raises(R.RpdeError):
        nav2.read(str(tmp_path / "missing.h5"))


def test_emu_adjoint_starts_from_a_navier2d_snapshot(emu_lib, tmp_path):
    """examples/navier_rbc_steady.rs: `navier.read_unwrap("restart.h5")` of a file Navier2D wrote, also at another
    resolution (field/io.rs:151-176), then descent; the oracle starts from the same coefficients."""
    src = R.Navier2D.new_confined(17, 17, 1e4, 1.0, 0.01, 1.0, "rbc", library=emu_lib, init_random=None)
    src.set_velocity(0.1, 1.0, 1.0)
    src.set_temperature(0.1, 1.0, 1.0)
    src.update(5)
    fn = str(tmp_path / "restart.h5")
    src.write(fn)
    nav = R.Navier2DAdjoint.new_confined(33, 33, 1e4, 1.0, 0.005, 1.0, "rbc", library=emu_lib)
    nav.read_unwrap(fn)
    nav.reset_time()
    old = src.temp.vhat
    assert np.array_equal(nav.temp.vhat[:old.shape[0], :old.shape[1]], old) and not nav.temp.vhat[old.shape[0]:].any()
    ora = A.Navier2DAdjoint.new_confined(33, 33, 1e4, 1.0, 0.005, 1.0, "rbc", eig_mode="parity")
    for k in ("velx", "vely", "temp"):
        getattr(ora, k).vhat = getattr(nav, k).vhat.copy()
    nav.update(3)
    for _ in range(3):
        ora.update()
    for k in ("velx", "vely", "temp", "velx_adj", "temp_adj"):
        assert rel(getattr(nav, k).vhat, getattr(ora, k).vhat) < 1e-10, k


def test_emu_adjoint_callback_writes_the_flow_file(emu_lib, tmp_path, monkeypatch, capfd):
    monkeypatch.chdir(tmp_path)
    nav = R.Navier2DAdjoint.new_confined(17, 17, 1e4, 1.0, 0.01, 1.0, "rbc", library=emu_lib)
    nav.set_velocity(0.1, 1.0, 1.0)
    nav.update(3)
    nav.callback()
    assert (tmp_path / "data" / "adjoint00000.03.h5").exists()
    out = capfd.readouterr().out
    assert "|div| =" in out and "|U| =" in out and "|T| =" in out


# ================================================================================================ Navier2DLnse
def check_lnse_parity(lib, nx, ny, periodic, steps, ra=1e5, dt=0.01, mean_flow=True, tol=1e-10, tol_p=1e-8, adjoint=False, bc="rbc"):
    """Engine vs oracle after every update, with a non-trivial mean flow (a convection roll) on top of the conduction
    profile: u, v, T to `tol`, pres / pseu to `tol_p` (the Poisson solve's amplified eigenvector round-off)."""
    from oracle import lnse as L
    mk_e = R.Navier2DLnse.new_periodic if periodic else R.Navier2DLnse.new_confined
    mk_o = L.Navier2DLnse.new_periodic if periodic else L.Navier2DLnse.new_confined
    nav = mk_e(nx, ny, ra, 1.0, dt, 1.0, bc, library=lib, mean_file="/nonexistent/mean.h5")
    ora = mk_o(nx, ny, ra, 1.0, dt, 1.0, bc, eig_mode="parity")
    assert rel(nav.mean_temp.v, ora.mean.temp.v) < 1e-13                  # the default mean: conduction profile ("hc": the parabolas of meanfield.rs:52-86), no flow
    assert np.abs(nav.mean_velx.v).max() == 0.0
    if mean_flow:
        x, y = ora.velx.x
        xs, ys = (x - x[0]) / (x[-1] - x[0]), (y - y[0]) / (y[-1] - y[0])
        um = 0.3 * np.sin(np.pi * xs)[:, None] * np.cos(np.pi * ys)[None, :]
        vm = -0.3 * np.cos(np.pi * xs)[:, None] * np.sin(np.pi * ys)[None, :]
        tm = ora.mean.temp.v + 0.1 * np.cos(np.pi * xs)[:, None] * np.sin(np.pi * ys)[None, :]
        for name, arr in (("velx", um), ("vely", vm), ("temp", tm)):
            getattr(nav, "mean_" + name).v = arr
            ora.mean.set_physical(name, arr)
        assert rel(nav.mean_velx.v, ora.mean.velx.v) < 1e-12
    for z in (nav, ora):
        z.set_velocity(0.1, 2.0, 1.0)
        z.set_temperature(0.1, 1.0, 2.0)
    worst = {}
    for s in range(steps):
        if adjoint:                          # Navier2DLnse::update_adjoint (lnse_adj_grad.rs:71-99)
            nav.update_adjoint(1)
            ora.update_adjoint()
        else:
            nav.update(1)
            ora.update()
        got, want = nav.spectral_fields(), ora.spectral_fields()
        for k in want:
            e = rel(got[k], want[k])
            worst[k] = max(worst.get(k, 0.0), e)
            assert e < (tol_p if k in ("pres", "pseu") else tol), (nx, ny, periodic, s, k, e)
    assert abs(nav.div_norm() - ora.div_norm()) < 1e-8 * max(ora.div_norm(), 1e-12)
    assert nav.exit() == ora.exit() and abs(nav.get_time() - ora.time) < 1e-12
    print("lnse adjoint" if adjoint else "lnse", nx, ny, "periodic" if periodic else "confined", {k: f"{v:.1e}" for k, v in worst.items()})
    return worst


def lnse_pair(lib, nx, ny, periodic, ra, pr, dt, seed=0, amp=1e-3, mean_flow=False):
    from oracle import lnse as L
    mk_e = R.Navier2DLnse.new_periodic if periodic else R.Navier2DLnse.new_confined
    mk_o = L.Navier2DLnse.new_periodic if periodic else L.Navier2DLnse.new_confined
    nav = mk_e(nx, ny, ra, pr, dt, 1.0, "rbc", library=lib, mean_file="/nonexistent/mean.h5")
    ora = mk_o(nx, ny, ra, pr, dt, 1.0, "rbc", eig_mode="parity")
    if mean_flow:
        x, y = ora.velx.x
        xs, ys = (x - x[0]) / (x[-1] - x[0]), (y - y[0]) / (y[-1] - y[0])
        for name, arr in (("velx", 0.3 * np.sin(np.pi * xs)[:, None] * np.cos(np.pi * ys)[None, :]),
                          ("vely", -0.3 * np.cos(np.pi * xs)[:, None] * np.sin(np.pi * ys)[None, :]),
                          ("temp", ora.mean.temp.v + 0.1 * np.cos(np.pi * xs)[:, None] * np.sin(np.pi * ys)[None, :])):
            getattr(nav, "mean_" + name).v = arr
            ora.mean.set_physical(name, arr)
    ora.init_random(amp, seed)               # the same random physical arrays on both sides
    for k in ("temp", "velx", "vely"):
        getattr(nav, k).v = getattr(ora, k).v
    return nav, ora


def check_lnse_gradient(lib, nx, ny, periodic, max_time, ra=3e3, pr=0.1, dt=0.01, with_target=False, fd_points=6, tol=1e-9, tmp_path=None):
    """grad_adjoint (fun_val and the three gradient fields), energy and grad_fd (a few points) of the engine against the oracle,
    the parameters of examples/navier_lnse_test_gradient.rs by default."""
    from oracle import lnse as L
    nav, ora = lnse_pair(lib, nx, ny, periodic, ra, pr, dt, mean_flow=with_target)
    target = None
    if with_target:                          # a target flow (lnse_adj_grad.rs:141-155, 160-166): MeanFields in the orthonormal space
        target = L.MeanFields(ora.field.space)
        rng = np.random.default_rng(7)
        x, y = ora.velx.x
        ys = (y - y[0]) / (y[-1] - y[0])
        for name in ("velx", "vely", "temp"):
            target.set_physical(name, 1e-3 * rng.standard_normal((nx, ny)) * np.sin(np.pi * ys)[None, :])
            getattr(target, name).backward()
    base = {k: getattr(ora, k).vhat.copy() for k in ("velx", "vely", "temp")}
    assert abs(nav.energy(0.5, 0.25) - ora.energy(0.5, 0.25)) < 1e-12 * ora.energy(0.5, 0.25)
    fn = None if tmp_path is None else str(tmp_path / "data" / "grad_adjoint.h5")
    fun_e, g_e = nav.grad_adjoint(max_time, None, 0.5, 0.25, target, filename=fn)
    fun_o, g_o = ora.grad_adjoint(max_time, 0.5, 0.25, target)
    assert abs(fun_e - fun_o) < tol * abs(fun_o), (fun_e, fun_o)
    errs = [rel(a, b) for a, b in zip(g_e, g_o)]
    assert max(errs) < tol, errs
    assert abs(nav.get_time() - ora.time) < 1e-12
    for k in ("velx", "vely", "temp"):       # both sides are left holding the adjoint fields
        assert rel(getattr(nav, k).vhat, getattr(ora, k).vhat) < tol, k
    if fn is not None:                       # the reference's data/grad_adjoint.h5: groups ux, uy, temp with v and vhat = forward(v)
        from tests.h5classic import File
        d = File(fn).datasets
        for g, arr, f in (("ux", g_e[0], ora.velx), ("uy", g_e[1], ora.vely), ("temp", g_e[2], ora.temp)):
            assert np.array_equal(d[g + "/v"], arr)
            want = f.space.forward(arr)
            got = d[g + "/vhat_re"] + 1j * d[g + "/vhat_im"] if periodic else d[g + "/vhat"]
            assert rel(got, want) < 1e-11, g
    # finite differences at a few points: restore the initial state on both sides first
    rng = np.random.default_rng(3)
    pts = [(("velx", "vely", "temp")[int(rng.integers(3))], int(rng.integers(nx)), int(rng.integers(1, ny - 1))) for _ in range(fd_points)]
    for z in (nav, ora):
        for k in base:
            getattr(z, k).vhat = base[k]
    for k in base:
        getattr(ora, k).backward()           # the reference perturbs the physical arrays the state holds (lnse_fd_grad.rs:41-43)
    fd_e = nav.grad_fd(max_time, None, 0.5, 0.25, points=pts, filename=None)
    fd_o = ora.grad_fd(max_time, 0.5, 0.25, points=pts)
    for a, b, k in zip(fd_e, fd_o, ("velx", "vely", "temp")):
        # differences of energies of 1e-7 divided by eps = 1e-5: round-off of 1e-16 relative in the energies is 1e-11 / |g|
        assert np.abs(a - b).max() < 1e-6 * max(np.abs(b).max(), 1e-30) + 1e-12, (k, np.abs(a - b).max(), np.abs(b).max())
    print("lnse gradient", nx, ny, "periodic" if periodic else "confined", f"fun {fun_o:.6e}", [f"{e:.1e}" for e in errs])


def test_oracle_lnse_is_the_linearisation_of_navier2d():
    """What pins the oracle's LNSE step: it IS the linearisation of Navier2D::update (oracle/navier.py, itself pinned by the
    critical Rayleigh numbers) about the conduction state (mean: no flow, T = the lift).  With N_eps = Navier2D started from
    eps * q and N_0 = Navier2D started from rest (its lift alone: the hydrostatic part the projection absorbs), the divided
    difference (N_eps - N_0) / eps equals the LNSE evolution of q to 1e-4 relative, and the part of the mismatch that is the
    quadratic term halves when eps halves."""
    from oracle import lnse as L
    n, ra, dt, steps = 33, 1e5, 0.01, 5

    def navier(eps):
        nav = N.Navier2D.new_confined(n, n, ra, 1.0, dt, 1.0, "rbc", eig_mode="parity")
        nav.set_velocity(eps, 2.0, 1.0)
        nav.set_temperature(eps, 1.0, 2.0)
        for _ in range(steps):
            nav.update()
        return {k: getattr(nav, k).vhat.copy() for k in ("velx", "vely", "temp")}

    lin = L.Navier2DLnse.new_confined(n, n, ra, 1.0, dt, 1.0, "rbc", eig_mode="parity")
    lin.set_velocity(1.0, 2.0, 1.0)
    lin.set_temperature(1.0, 1.0, 2.0)
    for _ in range(steps):
        lin.update()
    base = navier(0.0)
    mism = []
    for eps in (1e-3, 5e-4):
        ne = navier(eps)
        mism.append({k: np.abs((ne[k] - base[k]) / eps - getattr(lin, k).vhat).max() / np.abs(getattr(lin, k).vhat).max() for k in base})
    # measured: temp 6.4e-5 -> 3.2e-5 (the quadratic term, linear in eps); velx / vely 1.5e-5 / 1.3e-5 at both eps -- the flow
    # Navier2D develops from rest (its lift's discrete hydrostatic imbalance, |u| ~ 1e-5) advects the perturbation, a term the
    # linearisation about an exactly resting mean does not have
    for k in mism[0]:
        assert mism[0][k] < 2e-4, (k, mism)
        assert mism[1][k] <= 1.01 * mism[0][k], (k, mism)
    assert 1.8 < mism[0]["temp"] / mism[1]["temp"] < 2.2, mism


@pytest.mark.parametrize("nx,ny,periodic", [(33, 33, False), (65, 33, False), (32, 33, True)])
def test_emu_lnse_step_parity(emu_lib, nx, ny, periodic):
    check_lnse_parity(emu_lib, nx, ny, periodic, steps=4)


@pytest.mark.parametrize("nx,ny,periodic", [(33, 257, False), (32, 257, True)])
def test_emu_lnse_step_on_the_fused_schedule(emu_lib, monkeypatch, nx, ny, periodic):
    """Round 6: Navier2DLnse::update (lnse.rs:263-288) on Navier2DEngine's fused schedule -- no lift, the convection terms linearised
    about the mean fields inside the whole-line convection kernel (conv_line<N, true>: U d/dx f + V d/dy f + u d/dx M + v d/dy M,
    lnse_eq.rs:59-110), used where that kernel covers the y-lines (here: 257 points, emulation only).  Against the oracle after
    every update with a convection roll as mean flow, and against the composition of generic operators it replaces
    (RPDE_LNSE_FUSED=0): equal to round-off but not bit-identical -- the two forms really are different code."""
    monkeypatch.delenv("RPDE_LNSE_FUSED", raising=False)
    check_lnse_parity(emu_lib, nx, ny, periodic, steps=3)

    def run(flag):
        if flag is None:
            monkeypatch.delenv("RPDE_LNSE_FUSED", raising=False)
        else:
            monkeypatch.setenv("RPDE_LNSE_FUSED", flag)
        nav, _ = lnse_pair(emu_lib, nx, ny, periodic, 1e5, 1.0, 0.01, mean_flow=True)
        nav.update(2)
        nav.update(1)                          # a second hand-over of the state
        return nav.spectral_fields()
    fused, generic = run(None), run("0")
    monkeypatch.delenv("RPDE_LNSE_FUSED", raising=False)
    differs = False
    for k in generic:
        e = rel(fused[k], generic[k])
        assert e < (1e-8 if k in ("pres", "pseu") else 1e-11), (k, e)
        differs = differs or e > 0.0
    assert differs, "RPDE_LNSE_FUSED made no difference: the fused schedule did not run"


@pytest.mark.parametrize("nx,ny,periodic", [(17, 257, False), (16, 257, True)])
def test_emu_lnse_adjoint_step_on_the_fused_schedule(emu_lib, tmp_path, monkeypatch, nx, ny, periodic):
    """Round 6: Navier2DLnse::update_adjoint (lnse_adj_grad.rs:71-99, lnse_adj_eq.rs), confined and periodic, on Navier2DEngine's fused schedule:
    the convection terms -(U d/dx f + V d/dy f) + u* d_j U + v* d_j V + T* d_j T in the whole-line kernel (conv_line<N, 3>, the physical
    T* from a third y transform), no buoyancy in the vely equation, dt vely.to_ortho() of the step's start in the temperature equation.
    Against the oracle after every adjoint step, against the generic composition (round-off, not bit for bit), and through
    grad_adjoint (forward loop and adjoint loop fused, the gradient read from the physical arrays of the start of the last step)."""
    monkeypatch.delenv("RPDE_LNSE_FUSED", raising=False)
    check_lnse_parity(emu_lib, nx, ny, periodic, steps=3, adjoint=True)

    def run(flag):
        if flag is None:
            monkeypatch.delenv("RPDE_LNSE_FUSED", raising=False)
        else:
            monkeypatch.setenv("RPDE_LNSE_FUSED", flag)
        nav, _ = lnse_pair(emu_lib, nx, ny, periodic, 1e5, 1.0, 0.01, mean_flow=True)
        nav.update_adjoint(2)                  # several steps in one call: the last one starts from the state after the others
        nav.update_adjoint(1)
        return nav.spectral_fields(), {k: getattr(nav, k).v.copy() for k in ("velx", "vely", "temp")}
    (fused, fphys), (generic, gphys) = run(None), run("0")
    monkeypatch.delenv("RPDE_LNSE_FUSED", raising=False)
    differs = False
    for k in generic:
        e = rel(fused[k], generic[k])
        assert e < (1e-8 if k in ("pres", "pseu") else 1e-11), (k, e)
        differs = differs or e > 0.0
    assert differs, "RPDE_LNSE_FUSED made no difference: the fused schedule did not run"
    for k in gphys:                            # the physical arrays the state holds: those of the start of the last step, in both forms
        assert rel(fphys[k], gphys[k]) < 1e-11, k
    check_lnse_gradient(emu_lib, nx, ny, periodic, max_time=0.05, with_target=True, fd_points=2, tmp_path=tmp_path)


def test_emu_lnse_fused_schedule_follows_a_changed_mean(emu_lib, monkeypatch):
    """The fused schedule keeps the mean velocities and the mean gradients as device arrays of its own (set_lnse_mean_device): a mean
    field set BETWEEN two updates must reach them (GenericFlow2D::const_gen_).  Fused against generic with the mean flow switched on
    after the first update; and the state set from outside between updates (set_field) is what the next fused update starts from."""
    nx, ny = 17, 257

    def run(flag):
        if flag is None:
            monkeypatch.delenv("RPDE_LNSE_FUSED", raising=False)
        else:
            monkeypatch.setenv("RPDE_LNSE_FUSED", flag)
        nav, ora = lnse_pair(emu_lib, nx, ny, False, 1e5, 1.0, 0.01, mean_flow=False)
        nav.update(1)
        x, y = ora.velx.x
        xs, ys = (x - x[0]) / (x[-1] - x[0]), (y - y[0]) / (y[-1] - y[0])
        nav.mean_velx.v = 0.3 * np.sin(np.pi * xs)[:, None] * np.cos(np.pi * ys)[None, :]
        nav.mean_vely.v = -0.3 * np.cos(np.pi * xs)[:, None] * np.sin(np.pi * ys)[None, :]
        nav.update(1)
        nav.temp.v = 2.0 * nav.temp.v            # the state changed from outside
        nav.update(1)
        return nav.spectral_fields()
    fused, generic = run(None), run("0")
    monkeypatch.delenv("RPDE_LNSE_FUSED", raising=False)
    for k in generic:
        e = rel(fused[k], generic[k])
        assert e < (1e-8 if k in ("pres", "pseu") else 1e-11), (k, e)


@pytest.mark.parametrize("nx,ny,periodic", [(33, 33, False), (32, 33, True), (65, 257, False)])
def test_emu_adjoint_fused_forward_step(emu_lib, monkeypatch, nx, ny, periodic):
    """Round 6: the forward Navier-Stokes step inside Navier2DAdjoint::update (steady_adjoint.rs:547-585 -- Navier2D::update with
    DT_NAVIER and the buoyancy without the lift) runs on Navier2DEngine's fused schedule.  Against the composition of generic
    operators it replaces (RPDE_ADJOINT_FUSED=0) after three updates: all nine fields; both forms meet the oracle in
    check_step_parity of this module (the default form in every other test here)."""
    def run(flag):
        if flag is None:
            monkeypatch.delenv("RPDE_ADJOINT_FUSED", raising=False)
        else:
            monkeypatch.setenv("RPDE_ADJOINT_FUSED", flag)
        mk = R.Navier2DAdjoint.new_periodic if periodic else R.Navier2DAdjoint.new_confined
        nav = mk(nx, ny, 1e4, 1.0, 0.005, 1.0, "rbc", library=emu_lib)
        nav.set_velocity(0.2, 1.0, 1.0)
        nav.set_temperature(0.2, 1.0, 1.0)
        nav.update(3)
        return nav.spectral_fields()
    fused, generic = run(None), run("0")
    monkeypatch.delenv("RPDE_ADJOINT_FUSED", raising=False)
    for k in generic:
        assert rel(fused[k], generic[k]) < (1e-8 if k in ("pres", "pseu", "pres_adj") else 1e-11), (k, rel(fused[k], generic[k]))


@pytest.mark.parametrize("nx,ny,periodic", [(33, 33, False), (32, 33, True), (24, 25, False)])
def test_emu_lnse_hc(emu_lib, nx, ny, periodic):
    """bc = "hc" (lnse.rs:115-119, 202-206; nonlin.rs:117-121, 208-212; MeanFields::new_hc_*, meanfield.rs:52-86, 154-188): the
    temperature on cheb_dirichlet_neumann along y, the parabolic default mean -- forward and adjoint LNSE step and the non-linear
    solver against the oracle; Navier2DAdjoint keeps refusing it by name (steady_adjoint.rs:312-318)."""
    check_lnse_parity(emu_lib, nx, ny, periodic, steps=3, bc="hc")
    check_lnse_parity(emu_lib, nx, ny, periodic, steps=3, adjoint=True, mean_flow=False, bc="hc")
    check_nonlin_parity(emu_lib, nx, ny, periodic, steps=3, bc="hc")
    with pytest.raises(R.RpdeError, match="not supported by Navier2DAdjoint"):
        R.Navier2DAdjoint.new_confined(17, 17, 1e4, 1.0, 0.01, 1.0, "hc", library=emu_lib)


@pytest.mark.parametrize("nx,ny,periodic", [(33, 33, False), (32, 33, True), (16, 13, True)])
def test_emu_lnse_adjoint_step_parity(emu_lib, nx, ny, periodic):
    check_lnse_parity(emu_lib, nx, ny, periodic, steps=4, adjoint=True)


@pytest.mark.parametrize("nx,ny,periodic,with_target", [(16, 13, True, False), (17, 17, False, True)])
def test_emu_lnse_gradient_parity(emu_lib, tmp_path, nx, ny, periodic, with_target):
    check_lnse_gradient(emu_lib, nx, ny, periodic, max_time=0.1, with_target=with_target, tmp_path=tmp_path)


def test_oracle_lnse_adjoint_gradient_agrees_with_finite_differences():
    """What pins the restated adjoint equations (oracle/lnse.py <- lnse_adj_eq.rs, lnse_adj_grad.rs): the reference's own
    validation, examples/navier_lnse_test_gradient.rs -- adjoint gradient against finite differences over every grid point,
    accepted at |g_fd - g_adj| / |g_adj| <= 0.3.  The full example (horizon 10, 702 integrations of 1000 steps) ran once through
    tests/golden/make_lnse_gradient_pin.py and its result is committed; here: the committed record is inside the reference's
    acceptance, and a sub-sampled rerun (horizon 0.5, 40 points) reproduces the agreement."""
    import json
    from oracle import lnse as L
    path = os.path.join(os.path.dirname(__file__), "golden", "lnse_gradient_pin.json")
    rec = json.load(open(path))
    assert (rec["nx"], rec["ny"], rec["ra"], rec["pr"], rec["dt"], rec["max_time"]) == (18, 13, 3e3, 0.1, 0.01, 10.0)
    for k in ("velx", "vely", "temp"):
        assert rec[k]["rel_diff"] < rec["reference_acceptance"], (k, rec[k])
    nav = L.Navier2DLnse.new_periodic(18, 13, 3e3, 0.1, 0.01, 1.0, "rbc")
    nav.init_random(1e-3, 0)
    base = {k: getattr(nav, k).vhat.copy() for k in ("velx", "vely", "temp")}
    _, g_adj = nav.grad_adjoint(0.5, 0.5, 0.5)
    for k in base:
        getattr(nav, k).vhat = base[k]
        getattr(nav, k).backward()
    rng = np.random.default_rng(11)
    pts = [(k, int(rng.integers(18)), int(rng.integers(2, 11))) for k in ("velx", "vely", "temp") for _ in range(13)]
    g_fd = nav.grad_fd(0.5, 0.5, 0.5, points=pts)
    ga = np.array([-dict(zip(("velx", "vely", "temp"), g_adj))[k][i, j] for k, i, j in pts])
    gf = np.array([dict(zip(("velx", "vely", "temp"), g_fd))[k][i, j] for k, i, j in pts])
    assert np.linalg.norm(ga - gf) / np.linalg.norm(ga) < 0.3, (ga, gf)


# ------------------------------------------------------------------------------------------------ Navier2DNonLin
def _roll_mean(ora, nav=None, amp=0.3):
    x, y = ora.velx.x
    xs, ys = (x - x[0]) / (x[-1] - x[0]), (y - y[0]) / (y[-1] - y[0])
    for name, arr in (("velx", amp * np.sin(np.pi * xs)[:, None] * np.cos(np.pi * ys)[None, :]),
                      ("vely", -amp * np.cos(np.pi * xs)[:, None] * np.sin(np.pi * ys)[None, :]),
                      ("temp", ora.mean.temp.v + 0.1 * np.cos(np.pi * xs)[:, None] * np.sin(np.pi * ys)[None, :])):
        ora.mean.set_physical(name, arr)
        if nav is not None:
            getattr(nav, "mean_" + name).v = arr


@pytest.mark.parametrize("periodic,nx,ny", [(False, 17, 17), (True, 16, 17)])
def test_oracle_nonlin_about_the_conduction_state_is_navier2d(periodic, nx, ny):
    """What pins the restated non-linear equations (oracle/lnse.py Navier2DNonLin <- nonlin_eq.rs): about the default mean (no flow,
    the conduction profile) they ARE the equations of Navier2D (oracle/navier.py, pinned by the critical Rayleigh numbers) -- the
    mean temperature takes the lift's place in the buoyancy and in the convection term, its diffusion is the lift's.  Same initial
    state, same steps: equal to round-off."""
    from oracle import lnse as L
    mk_n = N.Navier2D.new_periodic if periodic else N.Navier2D.new_confined
    mk_l = L.Navier2DNonLin.new_periodic if periodic else L.Navier2DNonLin.new_confined
    nav = mk_n(nx, ny, 1e5, 1.0, 0.01, 1.0, "rbc", eig_mode="parity")
    nl = mk_l(nx, ny, 1e5, 1.0, 0.01, 1.0, "rbc", eig_mode="parity")
    for z in (nav, nl):
        z.set_velocity(0.2, 1.0, 1.0)
        z.set_temperature(0.2, 1.0, 1.0)
    for _ in range(10):
        nav.update()
        nl.update()
    for k in ("velx", "vely", "temp", "pres"):
        assert rel(getattr(nl, k).vhat, getattr(nav, k).vhat) < 1e-11, k


def test_oracle_nonlin_adjoint_history_terms_are_the_mean_terms():
    """nonlin_adj_eq.rs:31-48: the adjoint convection terms with the forward state have the form of the terms with the mean.  So the
    non-linear adjoint step about a mean M with the forward state H equals the LINEAR adjoint step (pinned by the reference's
    finite-difference example) about the mean M + H."""
    from oracle import lnse as L
    n = 17
    nl = L.Navier2DNonLin.new_confined(n, n, 1e4, 1.0, 0.01, 1.0, "rbc", eig_mode="parity")
    lin = L.Navier2DLnse.new_confined(n, n, 1e4, 1.0, 0.01, 1.0, "rbc", eig_mode="parity")
    _roll_mean(nl)
    nl.set_velocity(0.2, 2.0, 1.0)
    nl.set_temperature(0.1, 1.0, 2.0)
    for _ in range(3):
        nl.update_direct()
    h = nl.field_history[-1]
    for name in ("velx", "vely", "temp"):     # mean of the linear solver: M + H (physical arrays; H lies in the composite spaces)
        lin.mean.set_physical(name, getattr(nl.mean, name).space.backward(getattr(nl.mean, name).vhat) + h.v[name])
    rng = np.random.default_rng(4)
    for z in (nl, lin):
        z.time = 0.0
    for name in ("velx", "vely", "temp", "pres"):
        a = rng.standard_normal(getattr(nl, name).vhat.shape)
        getattr(nl, name).vhat = a.copy()
        getattr(lin, name).vhat = a.copy()
    nl.update_adjoint()
    lin.update_adjoint()
    for k in ("velx", "vely", "temp", "pres"):
        assert rel(getattr(nl, k).vhat, getattr(lin, k).vhat) < 1e-11, k
    assert len(nl.field_history) == 2


def check_nonlin_parity(lib, nx, ny, periodic, steps, ra=1e4, dt=0.01, tol=1e-10, tol_p=1e-8, max_time=None, tmp_path=None, bc="rbc"):
    """Navier2DNonLin: engine vs oracle through update_direct (with history), update_adjoint (consuming it), and grad_adjoint."""
    from oracle import lnse as L
    mk_e = R.Navier2DNonLin.new_periodic if periodic else R.Navier2DNonLin.new_confined
    mk_o = L.Navier2DNonLin.new_periodic if periodic else L.Navier2DNonLin.new_confined
    nav = mk_e(nx, ny, ra, 1.0, dt, 1.0, bc, library=lib, mean_file="/nonexistent/mean.h5")
    ora = mk_o(nx, ny, ra, 1.0, dt, 1.0, bc, eig_mode="parity")
    _roll_mean(ora, nav)
    for z in (nav, ora):
        z.set_velocity(0.2, 2.0, 1.0)
        z.set_temperature(0.1, 1.0, 2.0)
    base = {k: getattr(ora, k).vhat.copy() for k in ("velx", "vely", "temp")}
    worst = {}

    def compare(tag):
        got, want = nav.spectral_fields(), ora.spectral_fields()
        for k in want:
            e = rel(got[k], want[k])
            worst[k] = max(worst.get(k, 0.0), e)
            assert e < (tol_p if k in ("pres", "pseu") else tol), (nx, ny, periodic, tag, k, e)

    nav.update(1)                             # Integrate::update: no history entry (nonlin.rs:264-296)
    ora.update()
    compare("update")
    assert nav.field_history_len == 0
    for s in range(steps):
        nav.update_direct(1)
        ora.update_direct()
        compare(("direct", s))
    assert nav.field_history_len == steps == len(ora.field_history)
    for s in range(steps):
        nav.update_adjoint(1)
        ora.update_adjoint()
        compare(("adjoint", s))
    assert nav.field_history_len == 0
    with pytest.raises(R.RpdeError, match="history is empty"):
        nav.update_adjoint(1)
    if tmp_path is not None:                   # nonlin_io.rs:44-66: the snapshot carries the mean fields
        fn = str(tmp_path / "nl.h5")
        nav.write(fn)
        from tests.h5classic import File
        d = File(fn).datasets
        assert rel(d["ux_base/v"], nav.mean_velx.v) < 1e-14 and rel(d["temp_base/v"], nav.mean_temp.v) < 1e-14 and "uy_base/v" in d
    if max_time is not None:
        for z in (nav, ora):
            z.reset_time()
            for k in base:
                getattr(z, k).vhat = base[k]
            for k in ("pres", "pseu"):
                getattr(z, k).vhat = 0 * getattr(ora, k).vhat
        fun_e, g_e = nav.grad_adjoint(max_time, None, 0.5, 0.25, filename=None)
        fun_o, g_o = ora.grad_adjoint(max_time, 0.5, 0.25)
        assert abs(fun_e - fun_o) < 1e-9 * abs(fun_o)
        errs = [rel(a, b) for a, b in zip(g_e, g_o)]
        assert max(errs) < 1e-9, errs
        assert nav.field_history_len == 0
    print("nonlin", nx, ny, "periodic" if periodic else "confined", {k: f"{v:.1e}" for k, v in worst.items()})


@pytest.mark.parametrize("nx,ny,periodic", [(17, 257, False), (16, 257, True)])
def test_emu_nonlin_step_on_the_fused_schedule(emu_lib, tmp_path, monkeypatch, nx, ny, periodic):
    """Round 6: Navier2DNonLin::update (nonlin.rs:264-296) on Navier2DEngine's fused schedule where the whole-line convection kernel
    covers the y-lines -- the classic term with the mean velocities added to u, v and the mean gradients in the lift's place
    (conv_line<N, 2>), and what the mean fields add to the right-hand sides (their diffusion, the mean temperature in the buoyancy) as
    H^-1 of constant rows added behind the Helmholtz solves.  Against the oracle through update_direct, the adjoint steps that read
    the history, and grad_adjoint; and against the generic composition (RPDE_LNSE_FUSED=0): equal to round-off, not bit for bit."""
    monkeypatch.delenv("RPDE_LNSE_FUSED", raising=False)
    check_nonlin_parity(emu_lib, nx, ny, periodic, steps=3, max_time=0.05, tmp_path=tmp_path)
    from oracle import lnse as L

    def run(flag):
        if flag is None:
            monkeypatch.delenv("RPDE_LNSE_FUSED", raising=False)
        else:
            monkeypatch.setenv("RPDE_LNSE_FUSED", flag)
        mk_e = R.Navier2DNonLin.new_periodic if periodic else R.Navier2DNonLin.new_confined
        mk_o = L.Navier2DNonLin.new_periodic if periodic else L.Navier2DNonLin.new_confined
        nav = mk_e(nx, ny, 1e4, 1.0, 0.01, 1.0, "rbc", library=emu_lib, mean_file="/nonexistent/mean.h5")
        _roll_mean(mk_o(nx, ny, 1e4, 1.0, 0.01, 1.0, "rbc", eig_mode="parity"), nav)
        nav.set_velocity(0.2, 2.0, 1.0)
        nav.set_temperature(0.1, 1.0, 2.0)
        nav.update(2)
        nav.update(1)
        return nav.spectral_fields()
    fused, generic = run(None), run("0")
    monkeypatch.delenv("RPDE_LNSE_FUSED", raising=False)
    differs = False
    for k in generic:
        e = rel(fused[k], generic[k])
        assert e < (1e-8 if k in ("pres", "pseu") else 1e-11), (k, e)
        differs = differs or e > 0.0
    assert differs, "RPDE_LNSE_FUSED made no difference: the fused schedule did not run"


@pytest.mark.parametrize("nx,ny,periodic", [(17, 17, False), (33, 17, False), (16, 17, True)])
def test_emu_nonlin_parity(emu_lib, tmp_path, nx, ny, periodic):
    check_nonlin_parity(emu_lib, nx, ny, periodic, steps=3, max_time=0.05, tmp_path=tmp_path)


def check_lnse_callbacks(lib, tmp_path, monkeypatch, capfd, nonlinear):
    """Diagnostics against the oracle, the callback's files and lines, and grad_adjoint with `save_intervall`: the snapshots of the
    two loops appear under data/, and a snapshot at the last adjoint step refreshes the physical arrays the gradient is read from
    (lnse_io.rs:44-47) -- the oracle mirrors that, so the gradients still agree."""
    from oracle import lnse as L
    monkeypatch.chdir(tmp_path)
    nx, ny, dt = 17, 17, 0.05
    mk_e = R.Navier2DNonLin if nonlinear else R.Navier2DLnse
    mk_o = L.Navier2DNonLin if nonlinear else L.Navier2DLnse
    nav = mk_e.new_confined(nx, ny, 1e4, 1.0, dt, 1.0, "rbc", library=lib, mean_file="/nonexistent/mean.h5")
    ora = mk_o.new_confined(nx, ny, 1e4, 1.0, dt, 1.0, "rbc", eig_mode="parity")
    _roll_mean(ora, nav)
    for z in (nav, ora):
        z.set_velocity(0.2, 2.0, 1.0)
        z.set_temperature(0.1, 1.0, 2.0)
    base = {k: getattr(ora, k).vhat.copy() for k in ("velx", "vely", "temp")}
    nav.update(2)
    ora.update(); ora.update()
    d = nav.diagnostics()
    u2, v2, t2 = ora.averages_of_squares()
    for got, want in ((d["u2"], u2), (d["v2"], v2), (d["t2"], t2), (d["div"], ora.div_norm())):
        assert abs(got - want) < 1e-10 * abs(want), (got, want)
    if nonlinear:
        for key, want in (("nu", ora.eval_nu()), ("nuvol", ora.eval_nuvol()), ("re", ora.eval_re())):
            assert abs(d[key] - want) < 1e-10 * abs(want), (key, d[key], want)
    else:
        assert all(np.isnan(d[k]) for k in ("nu", "nuvol", "re"))
    capfd.readouterr()
    nav.callback_from_filename("data/x.h5", "data/info.txt", False, 0.1)
    out = capfd.readouterr().out
    assert (tmp_path / "data" / "x.h5").exists() and "|div| =" in out and "t2 =" in out and (("Nu =" in out) == nonlinear)
    cols = open(tmp_path / "data" / "info.txt").read().split()
    assert len(cols) == (7 if nonlinear else 4) and abs(float(cols[0]) - 0.1) < 1e-12 and abs(float(cols[-1]) - t2) < 1e-10 * t2
    # grad_adjoint with snapshots: horizon 1.0, save interval 0.5 -> flow files at 0.5 and 1.0 (the output interval is 1: only the
    # one at 1.0 is written), adjoint file at 1.0 = the last adjoint step
    for z in (nav, ora):
        z.reset_time()
        for k in base:
            getattr(z, k).vhat = base[k]
        for k in ("pres", "pseu"):
            getattr(z, k).vhat = 0 * getattr(ora, k).vhat
    if nonlinear:
        nav.clear_field_history()
        ora.field_history = []
    fun_e, g_e = nav.grad_adjoint(1.0, 0.5, 0.5, 0.25, filename="data/grad_adjoint.h5")
    fun_o, g_o = ora.grad_adjoint(1.0, 0.5, 0.25, save_intervall=0.5)
    assert ora.snapshots == [("flow", pytest.approx(1.0)), ("adjoint", pytest.approx(1.0))]
    names = sorted(os.listdir(tmp_path / "data"))
    assert "flow00001.00.h5" in names and "adjoint00001.00.h5" in names and "grad_adjoint.h5" in names and "flow00000.50.h5" not in names
    assert abs(fun_e - fun_o) < 1e-9 * abs(fun_o)
    assert max(rel(a, b) for a, b in zip(g_e, g_o)) < 1e-9
    assert rel(-g_e[0], ora.velx.space.backward(ora.velx.vhat)) < 1e-9     # refreshed: the gradient IS the final adjoint state
    lines = open(tmp_path / "data" / "info_adjoint.txt").read().strip().splitlines()
    assert len(lines) == 2                                                  # adjoint loop: times 0.5 and 1.0
    assert os.path.exists(tmp_path / "data" / "info.txt") == True
    assert len(open(tmp_path / "data" / "info.txt").read().strip().splitlines()) == (3 if nonlinear else 1)   # LNSE suppresses the forward loop's lines


@pytest.mark.parametrize("nonlinear", [False, True])
def test_emu_lnse_callbacks_and_saved_gradient_loops(emu_lib, tmp_path, monkeypatch, capfd, nonlinear):
    check_lnse_callbacks(emu_lib, tmp_path, monkeypatch, capfd, nonlinear)


def test_emu_grad_fd_base_run_snapshot_series(emu_lib, tmp_path, monkeypatch):
    """`grad_fd(max_time, Some(save_intervall), ...)` (lnse_fd_grad.rs:35, 54): the BASE run -- and only it -- calls
    Integrate::callback on the interval (data/flow{time:0>8.2}.h5 + a line of data/info.txt per save, lnse.rs:298-302); the
    gradient itself is the one of `None`."""
    monkeypatch.chdir(tmp_path)
    nav, ora = lnse_pair(emu_lib, 16, 13, True, 3e3, 0.1, 0.01)
    base = {k: getattr(nav, k).vhat.copy() for k in ("velx", "vely", "temp")}
    pts = [("velx", 3, 4), ("temp", 7, 6)]
    g0 = nav.grad_fd(1.0, None, 0.5, 0.25, points=pts, filename=None)
    assert not os.path.exists(tmp_path / "data")
    for k in base:
        getattr(nav, k).vhat = base[k]
    g1 = nav.grad_fd(1.0, 0.5, 0.5, 0.25, points=pts, filename=None)
    names = sorted(os.listdir(tmp_path / "data"))
    # callbacks at t = 0.5 and 1.0 of ONE run (the perturbed runs write nothing): an info line each, the flow snapshot on the
    # write interval of callback_from_filename(.., None) = OUTPUT_INTERVALL = 1 (lnse.rs:21, lnse_io.rs:80-90)
    assert names == ["flow00001.00.h5", "info.txt"], names
    assert len(open(tmp_path / "data" / "info.txt").read().strip().splitlines()) == 2
    for a, b in zip(g0, g1):
        assert np.array_equal(a, b)
    from tests.h5classic import File
    d = File(str(tmp_path / "data" / "flow00001.00.h5")).datasets
    assert abs(float(np.ravel(d["time"])[0]) - 1.0) < 1e-9 and d["ux/v"].shape == (16, 13)


def test_emu_l2_norm_and_steepest_descent(emu_lib):
    """functions::l2_norm and opt_routines::steepest_descent_energy_constrained (host arrays) against the oracle; the rotated
    state keeps the energy of the old one (the point of the routine) and alpha > 2 pi is refused like the reference's assert."""
    from oracle import lnse as L
    rng = np.random.default_rng(2)
    u0, v0, t0, gu, gv, gt = (rng.standard_normal((18, 13)) for _ in range(6))
    assert abs(R.l2_norm(u0, gu, v0, gv, t0, gt, 0.5, 0.25, library=emu_lib) - L.l2_norm(u0, gu, v0, gv, t0, gt, 0.5, 0.25)) < 1e-13
    (un_o, vn_o, tn_o), (gu_o, gv_o, gt_o) = L.steepest_descent_energy_constrained(u0, v0, t0, gu, gv, gt, 0.5, 0.25, 0.3)
    gu_e, gv_e, gt_e = gu.copy(), gv.copy(), gt.copy()
    un, vn, tn = (np.empty((18, 13)) for _ in range(3))
    R.steepest_descent_energy_constrained(u0, v0, t0, gu_e, gv_e, gt_e, un, vn, tn, 0.5, 0.25, 0.3, library=emu_lib)
    for a, b in ((un, un_o), (vn, vn_o), (tn, tn_o), (gu_e, gu_o), (gv_e, gv_o), (gt_e, gt_o)):
        assert rel(a, b) < 1e-13
    e0 = L.l2_norm(u0, u0, v0, v0, t0, t0, 0.5, 0.25)
    assert abs(L.l2_norm(un, un, vn, vn, tn, tn, 0.5, 0.25) - e0) < 1e-12 * e0
    with pytest.raises(R.RpdeError, match="2 pi"):
        R.steepest_descent_energy_constrained(u0, v0, t0, gu_e, gv_e, gt_e, un, vn, tn, 0.5, 0.25, 7.0, library=emu_lib)
    with pytest.raises(R.RpdeError, match="C-contiguous"):
        R.steepest_descent_energy_constrained(u0, v0, t0, gu_e.T, gv_e, gt_e, un, vn, tn, 0.5, 0.25, 0.3, library=emu_lib)


def test_emu_lnse_mean_from_a_snapshot_and_errors(emu_lib, tmp_path):
    """MeanFields::read_from_confined (meanfield.rs:92-127, 237-259): ux/v, uy/v, temp/v + tempbc/v of a Navier2D snapshot."""
    src = R.Navier2D.new_confined(17, 17, 1e4, 1.0, 0.01, 1.0, "rbc", library=emu_lib, init_random=None)
    src.set_velocity(0.1, 1.0, 1.0)
    src.set_temperature(0.1, 1.0, 1.0)
    src.update(3)
    fn = str(tmp_path / "mean.h5")
    src.write(fn)
    nav = R.Navier2DLnse.new_confined(17, 17, 1e4, 1.0, 0.01, 1.0, "rbc", library=emu_lib, mean_file=fn)
    from tests.h5classic import File
    f = File(fn).datasets
    assert rel(nav.mean_velx.v, f["ux/v"]) < 1e-12 and rel(nav.mean_temp.v, f["temp/v"] + f["tempbc/v"]) < 1e-12
    nav.set_velocity(0.05, 1.0, 1.0)
    nav.update(2)
    assert np.isfinite(nav.div_norm()) and not nav.exit()
    with pytest.raises(R.RpdeError, match="shape differs"):
        R.Navier2DLnse.new_confined(33, 17, 1e4, 1.0, 0.01, 1.0, "rbc", library=emu_lib, mean_file=fn)
    with pytest.raises(R.RpdeError, match="not recognized"):     # lnse.rs:118 (`"hc"` itself is accepted since round 6: test_emu_lnse_hc)
        R.Navier2DLnse.new_confined(17, 17, 1e4, 1.0, 0.01, 1.0, "xx", library=emu_lib, mean_file="/nonexistent")
    with pytest.raises(R.RpdeError, match="velx, vely or temp"):
        R.Navier2DLnse._Mean(nav, "pres").v


@pytest.mark.gpu
@pytest.mark.parametrize("nx,ny,periodic,steps", [(129, 129, False, 5), (256, 129, True, 4), (1025, 1025, False, 2), (129, 2049, False, 2), (65, 4097, False, 2), (256, 1025, True, 3)])
def test_gpu_lnse_adjoint_step_parity(hip_lib, nx, ny, periodic, steps):
    if nx >= 1025:
        check_lnse_parity(hip_lib, nx, ny, periodic, steps, ra=1e7, dt=1e-3, adjoint=True)
    else:
        check_lnse_parity(hip_lib, nx, ny, periodic, steps, adjoint=True)


@pytest.mark.gpu
@pytest.mark.parametrize("nx,ny,periodic,steps,max_time", [(65, 65, False, 4, 0.05), (128, 65, True, 3, 0.05), (513, 257, False, 2, None),
                                                           (129, 1025, False, 3, 0.03), (128, 1025, True, 2, None), (65, 4097, False, 2, None)])
def test_gpu_nonlin_parity(hip_lib, tmp_path, nx, ny, periodic, steps, max_time):
    check_nonlin_parity(hip_lib, nx, ny, periodic, steps, max_time=max_time, tmp_path=tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("nonlinear", [False, True])
def test_gpu_lnse_callbacks_and_saved_gradient_loops(hip_lib, tmp_path, monkeypatch, capfd, nonlinear):
    check_lnse_callbacks(hip_lib, tmp_path, monkeypatch, capfd, nonlinear)


@pytest.mark.gpu
@pytest.mark.parametrize("nx,ny,periodic,with_target,max_time", [(16, 13, True, False, 0.2), (33, 33, False, True, 0.1), (64, 33, True, True, 0.1)])
def test_gpu_lnse_gradient_parity(hip_lib, tmp_path, nx, ny, periodic, with_target, max_time):
    check_lnse_gradient(hip_lib, nx, ny, periodic, max_time=max_time, with_target=with_target, fd_points=4, tmp_path=tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("nx,ny,periodic,steps", [(129, 129, False, 5), (256, 129, True, 4), (1025, 1025, False, 2), (256, 1025, True, 3),
                                                  (129, 2049, False, 2), (65, 4097, False, 2)])
def test_gpu_lnse_step_parity(hip_lib, nx, ny, periodic, steps):
    if nx >= 1025:
        check_lnse_parity(hip_lib, nx, ny, periodic, steps, ra=1e7, dt=1e-3)
    else:
        check_lnse_parity(hip_lib, nx, ny, periodic, steps)
