"""``Navier2D`` of rustpde -- CPU oracle (test infrastructure; see oracle/__init__.py).

Follows, in this order:
  constructors      ``src/navier_stokes/navier.rs:215-308`` (confined), ``336-428`` (periodic)
  BC lift           ``src/navier_stokes/boundary_conditions.rs:18-36, 143-161`` ("rbc"), ``96-134, 163-202`` ("hc")
  nu / ka           ``src/navier_stokes/functions.rs:12-21``
  IC helpers        ``src/navier_stokes/functions.rs:85-126``
  conv_term/dealias ``src/navier_stokes/functions.rs:56-82``
  equations         ``src/navier_stokes/navier_eq.rs`` (whole file)
  update()          ``src/navier_stokes/navier.rs:438-466``
  diagnostics       ``src/navier_stokes/functions.rs:146-233``, ``src/field/average.rs:26-59``
The operation sequence of ``update()`` is kept exactly as in the reference (13 two-dimensional
transforms, 18 gradients, ...) so that the CPU baseline times the reference's algorithm.
"""
from __future__ import annotations

import numpy as np

from . import bases as B
from .solver import HholtzAdi, Poisson


def get_nu(ra, pr, height):
    return np.sqrt(pr / (ra / height ** 3.0))


def get_ka(ra, pr, height):
    return np.sqrt(1.0 / ((ra / height ** 3.0) * pr))


class Field2:
    """``FieldBase`` (``field.rs:59-129``): space, v (physical), vhat (spectral), x, dx."""

    def __init__(self, space: B.Space2):
        self.space = space
        self.v = space.ndarray_physical()
        self.vhat = space.ndarray_spectral()
        self.x = space.coords()
        self.dx = [self._get_dx(x, not b.is_cheb) for x, b in zip(self.x, space.bases)]

    @staticmethod
    def _get_dx(x, periodic):
        if periodic:
            return np.full(len(x), x[2] - x[1])
        left = np.concatenate([[x[0]], (x[1:] + x[:-1]) / 2.0])
        right = np.concatenate([(x[1:] + x[:-1]) / 2.0, [x[-1]]])
        return right - left

    def scale(self, scale):
        for i, sc in enumerate(scale):
            self.x[i] = self.x[i] * sc
            self.dx[i] = self.dx[i] * sc

    def forward(self):
        self.vhat = self.space.forward(self.v)

    def backward(self):
        self.v = self.space.backward(self.vhat)

    def to_ortho(self):
        return self.space.to_ortho(self.vhat)

    def gradient(self, deriv, scale=None):
        return self.space.gradient(self.vhat, deriv, scale)

    # averages, field/average.rs:26-59
    def average_axis(self, axis):
        length = abs(self.x[axis][-1] - self.x[axis][0])
        w = self.dx[axis] / length
        return (self.v * (w[:, None] if axis == 0 else w[None, :])).sum(axis=axis)

    def average(self):
        length = abs(self.x[1][-1] - self.x[1][0])
        return float((self.average_axis(0) * self.dx[1] / length).sum())


def _bc_rbc(space):
    """Linear conduction profile T = +0.5 (bottom) ... -0.5 (top), ``boundary_conditions.rs:18-36``."""
    f = Field2(space)
    y = f.x[1]
    x1, x2 = y[0], y[-1]
    y1, y2 = 0.5, -0.5
    m = (y2 - y1) / (x2 - x1)
    n = (y1 * x2 - y2 * x1) / (x2 - x1)
    f.v[:, :] = (m * y + n)[None, :]
    f.forward()
    f.backward()
    return f


def _bc_hc(space):
    """Horizontal convection, ``boundary_conditions.rs:96-134`` (confined) / ``163-202`` (periodic): per x a parabola
    in y with its vertex (value 0, slope 0) at the top wall y[n-1] and the value -0.5 cos(2 pi (x - x0) / L) at the
    bottom wall y[0]."""
    f = Field2(space)
    x, y = f.x
    x0, length = x[0], x[-1] - x[0]
    f_x = -0.5 * np.cos(2.0 * np.pi * (x - x0) / length)
    a = f_x / (y[0] - y[-1]) ** 2
    f.v[:, :] = a[:, None] * ((y - y[-1]) ** 2)[None, :]
    f.forward()
    f.backward()
    return f


def _apply_sin_cos(field, amp, m, n):
    x, y = field.x
    x = (x - x[0]) / (x[-1] - x[0])
    y = (y - y[0]) / (y[-1] - y[0])
    field.v = amp * np.sin(np.pi * m * x)[:, None] * np.cos(np.pi * n * y)[None, :]
    field.forward()


def _apply_cos_sin(field, amp, m, n):
    x, y = field.x
    x = (x - x[0]) / (x[-1] - x[0])
    y = (y - y[0]) / (y[-1] - y[0])
    field.v = amp * np.cos(np.pi * m * x)[:, None] * np.sin(np.pi * n * y)[None, :]
    field.forward()


class Navier2D:
    """Oracle mirror of ``Navier2D<T, S>`` (confined: T = f64, periodic: T = Complex<f64>)."""

    def __init__(self, nx, ny, ra, pr, dt, aspect, bc, periodic, eig_mode="full", eig_override=None):
        if bc not in ("rbc", "hc"):
            raise ValueError(f"Boundary condition type {bc!r} not recognized!")
        self.bc = bc
        # temperature: Dirichlet at both walls ("rbc") or Dirichlet at the bottom, Neumann at the top ("hc",
        # navier.rs:245-249 / 366-370); the lift carries the inhomogeneous part
        temp_y = B.cheb_dirichlet if bc == "rbc" else B.cheb_dirichlet_neumann
        lift = _bc_rbc if bc == "rbc" else _bc_hc
        self.periodic = periodic
        self.nx, self.ny = nx, ny
        self.scale = [aspect, 1.0]
        scale = self.scale
        nu = get_nu(ra, pr, scale[1] * 2.0)
        ka = get_ka(ra, pr, scale[1] * 2.0)
        self.params = {"ra": ra, "pr": pr, "nu": nu, "ka": ka}
        S = B.Space2
        if periodic:
            bx = B.fourier_r2c
            self.velx = Field2(S(bx(nx), B.cheb_dirichlet(ny)))
            self.vely = Field2(S(bx(nx), B.cheb_dirichlet(ny)))
            self.temp = Field2(S(bx(nx), temp_y(ny)))
            self.tempbc = lift(S(bx(nx), B.chebyshev(ny)))
            self.pres = Field2(S(bx(nx), B.chebyshev(ny)))
            self.pseu = Field2(S(bx(nx), B.cheb_neumann(ny)))
            self.field = Field2(S(bx(nx), B.chebyshev(ny)))
        else:
            self.velx = Field2(S(B.cheb_dirichlet(nx), B.cheb_dirichlet(ny)))
            self.vely = Field2(S(B.cheb_dirichlet(nx), B.cheb_dirichlet(ny)))
            self.temp = Field2(S(B.cheb_neumann(nx), temp_y(ny)))
            self.tempbc = lift(S(B.chebyshev(nx), B.chebyshev(ny)))
            self.pres = Field2(S(B.chebyshev(nx), B.chebyshev(ny)))
            self.pseu = Field2(S(B.cheb_neumann(nx), B.cheb_neumann(ny)))
            self.field = Field2(S(B.chebyshev(nx), B.chebyshev(ny)))
        for f in (self.velx, self.vely, self.temp, self.pres):
            f.scale(scale)
        c_nu = [dt * nu / scale[0] ** 2, dt * nu / scale[1] ** 2]
        c_ka = [dt * ka / scale[0] ** 2, dt * ka / scale[1] ** 2]
        self.solver_hholtz = [HholtzAdi(self.velx.space, c_nu), HholtzAdi(self.vely.space, c_nu),
                              HholtzAdi(self.temp.space, c_ka)]
        self.solver_pres = Poisson(self.pseu.space, [1.0 / scale[0] ** 2, 1.0 / scale[1] ** 2],
                                   eig_mode=eig_mode, eig_override=eig_override)
        self.rhs = np.zeros(self.field.space.shape_spectral, dtype=self.field.space.spectral_dtype)
        self.time = 0.0
        self.dt = dt
        # NOTE: the reference constructor ends with init_random(0.1) (unseeded RNG,
        # navier.rs:305); parity runs overwrite it with set_velocity/set_temperature or
        # explicit fields.  The oracle starts from zero.

    @classmethod
    def new_confined(cls, nx, ny, ra, pr, dt, aspect, bc, **kw):
        return cls(nx, ny, ra, pr, dt, aspect, bc, periodic=False, **kw)

    @classmethod
    def new_periodic(cls, nx, ny, ra, pr, dt, aspect, bc, **kw):
        return cls(nx, ny, ra, pr, dt, aspect, bc, periodic=True, **kw)

    # ------------------------------------------------------------------ initial conditions
    def set_velocity(self, amp, m, n):
        _apply_sin_cos(self.velx, amp, m, n)
        _apply_cos_sin(self.vely, -amp, m, n)

    def set_temperature(self, amp, m, n):
        _apply_cos_sin(self.temp, -amp, m, n)

    def init_random(self, amp, seed):
        """Seeded stand-in for ``init_random`` (``navier.rs:173-182``): uniform(-amp, amp) drawn on
        the host in the order temp, velx, vely; the same arrays are fed to the engine."""
        rng = np.random.default_rng(seed)
        for f in (self.temp, self.velx, self.vely):
            f.v = rng.uniform(-amp, amp, size=f.v.shape)
            f.forward()

    def set_field_physical(self, name, v):
        f = getattr(self, name)
        f.v = np.array(v, dtype=np.float64, copy=True)
        f.forward()

    def reset_time(self):
        self.time = 0.0

    # ------------------------------------------------------------------ equations
    def zero_rhs(self):
        self.rhs = np.zeros_like(self.rhs)

    def div(self):
        self.zero_rhs()
        self.rhs = self.rhs + self.velx.gradient([1, 0], self.scale)
        self.rhs = self.rhs + self.vely.gradient([0, 1], self.scale)
        return self.rhs.copy()

    def div_norm(self):
        d = self.div()
        return float(np.sqrt((d.real ** 2 + d.imag ** 2).sum()))

    def _conv_term(self, u, field, deriv):
        return u * self.field.space.backward(field.gradient(deriv, self.scale))

    def _dealias(self, vhat):
        n_x = vhat.shape[0] * 2 // 3
        n_y = vhat.shape[1] * 2 // 3
        vhat[n_x:, :] = 0
        vhat[:, n_y:] = 0

    def _conv_finish(self, conv):
        self.field.v = conv
        self.field.forward()
        self._dealias(self.field.vhat)
        return self.field.vhat.copy()

    def conv_velx(self, ux, uy):
        conv = self._conv_term(ux, self.velx, [1, 0])
        conv += self._conv_term(uy, self.velx, [0, 1])
        return self._conv_finish(conv)

    def conv_vely(self, ux, uy):
        conv = self._conv_term(ux, self.vely, [1, 0])
        conv += self._conv_term(uy, self.vely, [0, 1])
        return self._conv_finish(conv)

    def conv_temp(self, ux, uy):
        conv = self._conv_term(ux, self.temp, [1, 0])
        conv += self._conv_term(uy, self.temp, [0, 1])
        conv += self._conv_term(ux, self.tempbc, [1, 0])
        conv += self._conv_term(uy, self.tempbc, [0, 1])
        return self._conv_finish(conv)

    def solve_velx(self, ux, uy):
        self.zero_rhs()
        self.rhs += self.velx.to_ortho()
        self.rhs -= self.pres.gradient([1, 0], self.scale) * self.dt
        self.rhs -= self.conv_velx(ux, uy) * self.dt
        self.velx.vhat = self.solver_hholtz[0].solve(self.rhs)

    def solve_vely(self, ux, uy, buoy):
        self.zero_rhs()
        self.rhs += self.vely.to_ortho()
        self.rhs -= self.pres.gradient([0, 1], self.scale) * self.dt
        self.rhs += buoy * self.dt
        self.rhs -= self.conv_vely(ux, uy) * self.dt
        self.vely.vhat = self.solver_hholtz[1].solve(self.rhs)

    def solve_temp(self, ux, uy):
        self.zero_rhs()
        self.rhs += self.temp.to_ortho()
        ka = self.params["ka"]
        self.rhs += self.tempbc.gradient([2, 0], self.scale) * self.dt * ka
        self.rhs += self.tempbc.gradient([0, 2], self.scale) * self.dt * ka
        self.rhs -= self.conv_temp(ux, uy) * self.dt
        self.temp.vhat = self.solver_hholtz[2].solve(self.rhs)

    def solve_pres(self, f):
        self.pseu.vhat = self.solver_pres.solve(f)
        self.pseu.vhat[0, 0] = 0.0

    def correct_velocity(self, c):
        dp_dx = self.pseu.gradient([1, 0], self.scale) * (-c)
        dp_dy = self.pseu.gradient([0, 1], self.scale) * (-c)
        self.velx.vhat = self.velx.vhat + self.velx.space.from_ortho(dp_dx)
        self.vely.vhat = self.vely.vhat + self.vely.space.from_ortho(dp_dy)

    def update_pres(self, div):
        a = -1.0 * self.params["nu"]
        b = 1.0 / self.dt
        self.pres.vhat = self.pres.vhat + div * a + self.pseu.to_ortho() * b

    # ------------------------------------------------------------------ time step
    def update(self):
        """``Integrate::update`` (``navier.rs:438-466``)."""
        that = self.temp.to_ortho() + self.tempbc.to_ortho()
        self.velx.backward()
        self.vely.backward()
        ux = self.velx.v.copy()
        uy = self.vely.v.copy()
        self.solve_velx(ux, uy)
        self.solve_vely(ux, uy, that)
        div = self.div()
        self.solve_pres(div)
        self.correct_velocity(1.0)
        self.update_pres(div)
        self.solve_temp(ux, uy)
        self.time += self.dt

    def exit(self):
        return bool(np.isnan(self.div_norm()))

    def integrate(self, max_time, max_steps=10_000_000):
        """``integrate`` (``src/lib.rs:187-219``) without the I/O callback."""
        eps_dt = self.dt * 1e-4
        step = 0
        while True:
            self.update()
            step += 1
            if self.time + eps_dt >= max_time or step >= max_steps or self.exit():
                break
        return step

    # ------------------------------------------------------------------ outputs
    def physical_fields(self):
        """u, v, T (perturbation), p in physical space, as ``Navier2D::write`` produces them
        (``navier_io.rs:44-62`` calls ``backward`` on each field first)."""
        for f in (self.velx, self.vely, self.temp, self.pres):
            f.backward()
        return {"velx": self.velx.v.copy(), "vely": self.vely.v.copy(),
                "temp": self.temp.v.copy(), "pres": self.pres.v.copy()}

    # diagnostics, functions.rs:146-233
    def eval_nu(self):
        fld = self.field
        fld.vhat = self.temp.to_ortho() + self.tempbc.to_ortho()
        fld.vhat = fld.gradient([0, 1], None) * (-2.0 / self.scale[1])
        fld.backward()
        self._copy_grid(fld)
        x_avg = fld.average_axis(0)
        return float((x_avg[-1] + x_avg[0]) / 2.0)

    def eval_nuvol(self):
        fld = self.field
        ka = self.params["ka"]
        fld.vhat = self.temp.to_ortho() + self.tempbc.to_ortho()
        fld.backward()
        self.vely.backward()
        vely_temp = fld.v * self.vely.v
        fld.vhat = fld.gradient([0, 1], None) / (self.scale[1] * -1.0)
        fld.backward()
        fld.v = (fld.v + vely_temp / ka) * 2.0 * self.scale[1]
        self._copy_grid(fld)
        return fld.average()

    def eval_re(self):
        fld = self.field
        nu = self.params["nu"]
        self.velx.backward()
        self.vely.backward()
        fld.v = np.sqrt(self.velx.v ** 2 + self.vely.v ** 2) * (2.0 * self.scale[1] / nu)
        self._copy_grid(fld)
        return fld.average()

    def _copy_grid(self, fld):
        # `field` is never scaled in the reference (navier.rs:256-261): averages use the
        # unscaled grid of `field` itself; weights dx/length are scale invariant.
        return fld


class Statistics:
    """Oracle mirror of ``Statistics<T, S>`` (``src/navier_stokes/statistics.rs:11-108``): fields of the
    orthonormal ``field`` space -- ``t_avg`` is a running mean, ``ux_avg`` / ``uy_avg`` and ``nusselt`` hold the
    LAST snapshot (the reference assigns them, ``statistics.rs:98-103``)."""

    def __init__(self, navier: Navier2D, save_stat, write_stat):   # Statistics::new, statistics.rs:50-76
        self.params = dict(navier.params)
        self.scale = list(navier.scale)
        space = navier.field.space
        self.field = Field2(space)
        self.t_avg, self.ux_avg, self.uy_avg, self.nusselt = (Field2(space) for _ in range(4))
        self.save_stat, self.write_stat = save_stat, write_stat
        self.avg_time = 0.0
        self.tot_time = navier.time
        self.num_save = 0

    def update(self, that, uxhat, uyhat, time):                     # statistics.rs:84-108
        if time < self.tot_time:
            print(f"Statistics time mismatch (navier < stat): {time!r} < {self.tot_time!r}")
            return
        weight = float(self.num_save)
        self.t_avg.vhat = (self.t_avg.vhat * weight + that) / (weight + 1.0)
        self.ux_avg.vhat = np.array(uxhat, copy=True)
        self.uy_avg.vhat = np.array(uyhat, copy=True)
        self._nusselt(that, uyhat, self.params["ka"])
        self.nusselt.vhat = self.field.vhat.copy()
        self.num_save += 1
        self.avg_time += time - self.tot_time
        self.tot_time = time

    def _nusselt(self, that, uyhat, kappa):                         # fn nusselt, statistics.rs:248-271
        field = self.field
        field.vhat = np.array(uyhat, copy=True)
        field.backward()
        uy_v = field.v.copy()
        field.vhat = np.array(that, copy=True)
        field.backward()
        uy_temp = field.v * uy_v
        dtdz = field.gradient([0, 1], None) / (self.scale[1] * -1.0)
        field.vhat = dtdz
        field.backward()
        field.v = (field.v + uy_temp / kappa) * 2.0 * self.scale[1]
        field.forward()

    def update_from(self, navier: Navier2D):
        """The one call site of the reference (``navier_io.rs:110-115``)."""
        self.update(navier.temp.to_ortho(), navier.velx.to_ortho(), navier.vely.to_ortho(), navier.time)
