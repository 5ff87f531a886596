"""``Navier2DLnse`` of rustpde (Navier-Stokes linearised about mean fields) -- CPU oracle (test infrastructure; see
oracle/__init__.py).

Follows, in this order:
  mean fields       ``src/navier_stokes_lnse/meanfield.rs:20-56`` ("rbc" default), ``237-259`` (read: physical arrays + lift)
  constructors      ``src/navier_stokes_lnse/lnse.rs:98-176`` (confined), ``196-253`` (periodic)
  equations         ``src/navier_stokes_lnse/lnse_eq.rs`` (whole file)
  update() / exit() ``lnse.rs:263-288, 305-313``
The reference holds no golden output for this step; it is the step of ``Navier2D`` (oracle/navier.py, pinned by the two
critical Rayleigh numbers) with other convection terms, and the two are tied together by a test: linearised about a mean
flow M, ``Navier2D`` started from M + eps * q follows M(t) + eps * q_lnse(t) up to O(eps^2)
(tests/test_adjoint.py::test_oracle_lnse_is_the_linearisation_of_navier2d).
"""
from __future__ import annotations

import numpy as np

from . import bases as B
from .navier import Field2, _apply_cos_sin, _apply_sin_cos, get_ka, get_nu
from .solver import HholtzAdi, Poisson


class MeanFields:
    """``MeanFields`` (meanfield.rs:11-18): velx, vely, temp in orthonormal spaces."""

    def __init__(self, space):
        self.velx, self.vely, self.temp = Field2(space), Field2(space), Field2(space)

    @classmethod
    def new_rbc(cls, space):
        m = cls(space)
        y = m.temp.x[1]
        height = y[-1] - y[0]
        m.temp.v[:, :] = (-(y - y[0]) / height + 0.5)[None, :]
        m.temp.forward()
        return m

    def set_physical(self, name, v):
        f = getattr(self, name)
        f.v = np.array(v, dtype=np.float64, copy=True)
        f.forward()


class Navier2DLnse:
    def __init__(self, nx, ny, ra, pr, dt, aspect, bc, periodic, eig_mode="full"):
        if bc != "rbc":
            raise ValueError(f"Boundary condition type {bc!r} not supported by this oracle")
        self.periodic, self.nx, self.ny = periodic, nx, ny
        self.scale = scale = [aspect, 1.0]
        nu = get_nu(ra, pr, scale[1] * 2.0)
        ka = get_ka(ra, pr, scale[1] * 2.0)
        self.params = {"ra": ra, "pr": pr, "nu": nu, "ka": ka}
        S = B.Space2
        bx = B.fourier_r2c if periodic else None
        if periodic:
            self.field = Field2(S(bx(nx), B.chebyshev(ny)))
            self.velx = Field2(S(bx(nx), B.cheb_dirichlet(ny)))
            self.vely = Field2(S(bx(nx), B.cheb_dirichlet(ny)))
            self.pres = Field2(S(bx(nx), B.chebyshev(ny)))
            self.pseu = Field2(S(bx(nx), B.cheb_neumann(ny)))
            self.temp = Field2(S(bx(nx), B.cheb_dirichlet(ny)))
        else:
            self.field = Field2(S(B.chebyshev(nx), B.chebyshev(ny)))
            self.velx = Field2(S(B.cheb_dirichlet(nx), B.cheb_dirichlet(ny)))
            self.vely = Field2(S(B.cheb_dirichlet(nx), B.cheb_dirichlet(ny)))
            self.pres = Field2(S(B.chebyshev(nx), B.chebyshev(ny)))
            self.pseu = Field2(S(B.cheb_neumann(nx), B.cheb_neumann(ny)))
            self.temp = Field2(S(B.cheb_neumann(nx), B.cheb_dirichlet(ny)))
        for f in (self.velx, self.vely, self.temp, self.pres):
            f.scale(scale)
        self.mean = MeanFields.new_rbc(self.field.space)
        c_nu = [dt * nu / scale[0] ** 2, dt * nu / scale[1] ** 2]
        c_ka = [dt * ka / scale[0] ** 2, dt * ka / scale[1] ** 2]
        self.solver_hholtz = [HholtzAdi(self.velx.space, c_nu), HholtzAdi(self.vely.space, c_nu), HholtzAdi(self.temp.space, c_ka)]
        self.solver_pres = Poisson(self.pseu.space, [1.0 / scale[0] ** 2, 1.0 / scale[1] ** 2], eig_mode=eig_mode)
        self.rhs = np.zeros(self.field.space.shape_spectral, dtype=self.field.space.spectral_dtype)
        self.time, self.dt = 0.0, dt

    @classmethod
    def new_confined(cls, nx, ny, ra, pr, dt, aspect, bc, **kw):
        return cls(nx, ny, ra, pr, dt, aspect, bc, periodic=False, **kw)

    @classmethod
    def new_periodic(cls, nx, ny, ra, pr, dt, aspect, bc, **kw):
        return cls(nx, ny, ra, pr, dt, aspect, bc, periodic=True, **kw)

    def set_velocity(self, amp, m, n):
        _apply_sin_cos(self.velx, amp, m, n)
        _apply_cos_sin(self.vely, -amp, m, n)

    def set_temperature(self, amp, m, n):
        _apply_cos_sin(self.temp, -amp, m, n)

    # ------------------------------------------------------------------ lnse_eq.rs
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

    def _conv(self, ux, uy, mean_f, f):          # lnse_eq.rs:59-110
        self.mean.velx.backward()
        self.mean.vely.backward()
        um, vm = self.mean.velx.v, self.mean.vely.v
        conv = self._conv_term(ux, mean_f, [1, 0])
        conv += self._conv_term(uy, mean_f, [0, 1])
        conv += self._conv_term(um, f, [1, 0])
        conv += self._conv_term(vm, f, [0, 1])
        self.field.v = conv
        self.field.forward()
        vhat = self.field.vhat
        vhat[vhat.shape[0] * 2 // 3:, :] = 0
        vhat[:, vhat.shape[1] * 2 // 3:] = 0
        return vhat.copy()

    def solve_velx(self, ux, uy):
        self.zero_rhs()
        self.rhs += self.velx.to_ortho()
        self.rhs -= self.pres.gradient([1, 0], self.scale) * self.dt
        self.rhs -= self._conv(ux, uy, self.mean.velx, self.velx) * self.dt
        self.velx.vhat = self.solver_hholtz[0].solve(self.rhs)

    def solve_vely(self, ux, uy, buoy):
        self.zero_rhs()
        self.rhs += self.vely.to_ortho()
        self.rhs -= self.pres.gradient([0, 1], self.scale) * self.dt
        self.rhs += buoy * self.dt
        self.rhs -= self._conv(ux, uy, self.mean.vely, self.vely) * self.dt
        self.vely.vhat = self.solver_hholtz[1].solve(self.rhs)

    def solve_temp(self, ux, uy):
        self.zero_rhs()
        self.rhs += self.temp.to_ortho()
        self.rhs -= self._conv(ux, uy, self.mean.temp, self.temp) * self.dt
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
        self.pres.vhat = self.pres.vhat + div * (-1.0 * self.params["nu"]) + self.pseu.to_ortho() * (1.0 / self.dt)

    def update(self):                            # lnse.rs:263-288
        that = self.temp.to_ortho()
        self.velx.backward()
        self.vely.backward()
        ux, uy = self.velx.v.copy(), self.vely.v.copy()
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

    def spectral_fields(self, names=("velx", "vely", "temp", "pres", "pseu")):
        return {k: getattr(self, k).vhat.copy() for k in names}
