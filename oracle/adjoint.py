"""``Navier2DAdjoint`` of rustpde (adjoint descent to steady states, Farazmand 2016) -- CPU oracle
(test infrastructure; see oracle/__init__.py).

Follows, in this order:
  constants         ``src/navier_stokes/steady_adjoint.rs:59-64`` (RES_TOL, WEIGHT_LAPLACIAN, DT_NAVIER)
  constructors      ``steady_adjoint.rs:215-370`` (confined), ``372-531`` (periodic)
  equations         ``src/navier_stokes/steady_adjoint_eq.rs`` (whole file)
  update() / exit() ``steady_adjoint.rs:541-608, 624-638``
  write()           ``src/navier_stokes/steady_adjoint_io.rs:48-71`` (fields only)
The norm of the residual is the tensor Helmholtz solver ``Hholtz`` (``src/solver/hholtz.rs``; oracle/solver.py),
pinned by the reference's own analytic tests (tests/test_oracle_golden.py::test_hholtz_tensor_analytic).  The
reference holds no golden output for ``update()`` of this solver either: step parity is defined by this file.

Kept exactly as the reference has them, including what looks unintended there:
  * the forward step's buoyancy is ``temp.to_ortho() * dt`` WITHOUT the lift (``steady_adjoint_eq.rs:151``; ``Navier2D``
    adds ``tempbc``, ``navier.rs:441-444``);
  * ``conv_velx_adjoint`` adds ``conv_term(velx, velx_adj, [1, 0])`` twice (``steady_adjoint_eq.rs:245, 251``) where the
    comment announces ``-(dj Ui) ui*``; the commented-out lines are not evaluated;
  * the adjoint momentum and temperature equations are explicit: ``velx.from_ortho(&rhs)`` replaces the Helmholtz solve
    (``steady_adjoint_eq.rs:358, 389, 424``);
  * ``update_pres_adj`` ignores the divergence (``steady_adjoint_eq.rs:203-210``).
"""
from __future__ import annotations

import numpy as np

from . import bases as B
from .navier import Field2, _apply_cos_sin, _apply_sin_cos, _bc_hc, _bc_rbc, get_ka, get_nu
from .solver import Hholtz, HholtzAdi, Poisson

RES_TOL = 1e-7            # steady_adjoint.rs:60
WEIGHT_LAPLACIAN = 1e-1   # steady_adjoint.rs:62
DT_NAVIER = 1e-3          # steady_adjoint.rs:64


class Navier2DAdjoint:
    """Oracle mirror of ``Navier2DAdjoint<T, S>`` (confined: T = f64, periodic: T = Complex<f64>)."""

    def __init__(self, nx, ny, ra, pr, dt, aspect, bc, periodic, eig_mode="full"):
        if bc not in ("rbc", "hc"):
            raise ValueError(f"Boundary condition type {bc!r} not recognized!")
        self.bc, self.periodic, self.nx, self.ny = bc, periodic, nx, ny
        self.scale = scale = [aspect, 1.0]
        nu = get_nu(ra, pr, scale[1] * 2.0)
        ka = get_ka(ra, pr, scale[1] * 2.0)
        self.params = {"ra": ra, "pr": pr, "nu": nu, "ka": ka}
        temp_y = B.cheb_dirichlet if bc == "rbc" else B.cheb_dirichlet_neumann
        lift = _bc_rbc if bc == "rbc" else _bc_hc
        S = B.Space2
        if periodic:          # steady_adjoint.rs:388-420
            bx = B.fourier_r2c
            self.velx = Field2(S(bx(nx), B.cheb_dirichlet(ny)))
            self.vely = Field2(S(bx(nx), B.cheb_dirichlet(ny)))
            self.temp = Field2(S(bx(nx), temp_y(ny)))
            self.tempbc = lift(S(bx(nx), B.chebyshev(ny)))
            self.pres = Field2(S(bx(nx), B.chebyshev(ny)))
            self.pseu = Field2(S(bx(nx), B.cheb_neumann(ny)))
            self.field = Field2(S(bx(nx), B.chebyshev(ny)))
        else:                 # steady_adjoint.rs:236-259
            self.velx = Field2(S(B.cheb_dirichlet(nx), B.cheb_dirichlet(ny)))
            self.vely = Field2(S(B.cheb_dirichlet(nx), B.cheb_dirichlet(ny)))
            self.temp = Field2(S(B.cheb_neumann(nx), temp_y(ny)))
            self.tempbc = lift(S(B.chebyshev(nx), B.chebyshev(ny)))
            self.pres = Field2(S(B.chebyshev(nx), B.chebyshev(ny)))
            self.pseu = Field2(S(B.cheb_neumann(nx), B.cheb_neumann(ny)))
            self.field = Field2(S(B.chebyshev(nx), B.chebyshev(ny)))
        for f in (self.velx, self.vely, self.temp, self.pres):
            f.scale(scale)
        # adjoint fields: clones (steady_adjoint.rs:267-271)
        self.velx_adj = Field2(self.velx.space); self.velx_adj.scale(scale)
        self.vely_adj = Field2(self.vely.space); self.vely_adj.scale(scale)
        self.temp_adj = Field2(self.temp.space); self.temp_adj.scale(scale)
        self.pres_adj = Field2(self.pres.space); self.pres_adj.scale(scale)
        # Helmholtz solvers of the FORWARD step run on DT_NAVIER, not on dt (steady_adjoint.rs:273-295)
        c_nu = [DT_NAVIER * nu / scale[0] ** 2, DT_NAVIER * nu / scale[1] ** 2]
        c_ka = [DT_NAVIER * ka / scale[0] ** 2, DT_NAVIER * ka / scale[1] ** 2]
        self.solver_hholtz = [HholtzAdi(self.velx.space, c_nu), HholtzAdi(self.vely.space, c_nu),
                              HholtzAdi(self.temp.space, c_ka)]
        self.solver_pres = Poisson(self.pseu.space, [1.0 / scale[0] ** 2, 1.0 / scale[1] ** 2], eig_mode=eig_mode)
        # smoother (1 - weight * D2) (steady_adjoint.rs:300-322)
        c_w = [WEIGHT_LAPLACIAN / scale[0] ** 2, WEIGHT_LAPLACIAN / scale[1] ** 2]
        self.solver_norm = [Hholtz(self.velx.space, c_w, eig_mode=eig_mode), Hholtz(self.vely.space, c_w, eig_mode=eig_mode),
                            Hholtz(self.temp.space, c_w, eig_mode=eig_mode)]
        self.rhs = np.zeros(self.field.space.shape_spectral, dtype=self.field.space.spectral_dtype)
        self.time = 0.0
        self.dt = dt

    @classmethod
    def new_confined(cls, nx, ny, ra, pr, dt, aspect, bc, **kw):
        return cls(nx, ny, ra, pr, dt, aspect, bc, periodic=False, **kw)

    @classmethod
    def new_periodic(cls, nx, ny, ra, pr, dt, aspect, bc, **kw):
        return cls(nx, ny, ra, pr, dt, aspect, bc, periodic=True, **kw)

    # ------------------------------------------------------------------ initial conditions (steady_adjoint.rs:178-213)
    def set_velocity(self, amp, m, n):
        _apply_sin_cos(self.velx, amp, m, n)
        _apply_cos_sin(self.vely, -amp, m, n)

    def set_temperature(self, amp, m, n):
        _apply_cos_sin(self.temp, -amp, m, n)

    def set_field_physical(self, name, v):
        f = getattr(self, name)
        f.v = np.array(v, dtype=np.float64, copy=True)
        f.forward()

    def reset_time(self):
        self.time = 0.0

    # ------------------------------------------------------------------ general (steady_adjoint_eq.rs:18-68)
    def zero_rhs(self):
        self.rhs = np.zeros_like(self.rhs)

    def div(self):
        self.zero_rhs()
        self.rhs = self.rhs + self.velx.gradient([1, 0], self.scale)
        self.rhs = self.rhs + self.vely.gradient([0, 1], self.scale)
        return self.rhs.copy()

    @staticmethod
    def _norm_l2(a):
        return float(np.sqrt((a.real ** 2 + a.imag ** 2).sum()))

    def div_norm(self):
        return self._norm_l2(self.div())

    def norm_residual(self):
        return [self._norm_l2(self.velx_adj.vhat), self._norm_l2(self.vely_adj.vhat), self._norm_l2(self.temp_adj.vhat)]

    # ------------------------------------------------------------------ convection, forward (steady_adjoint_eq.rs:71-121)
    def _conv_term(self, u, field, deriv):       # functions.rs:56-69
        return u * self.field.space.backward(field.gradient(deriv, self.scale))

    def _conv_finish(self, conv):                # forward + dealias (functions.rs:72-82)
        self.field.v = conv
        self.field.forward()
        vhat = self.field.vhat
        vhat[vhat.shape[0] * 2 // 3:, :] = 0
        vhat[:, vhat.shape[1] * 2 // 3:] = 0
        return vhat.copy()

    def conv_temp(self, ux, uy):
        conv = self._conv_term(ux, self.temp, [1, 0])
        conv += self._conv_term(uy, self.temp, [0, 1])
        conv += self._conv_term(ux, self.tempbc, [1, 0])
        conv += self._conv_term(uy, self.tempbc, [0, 1])
        return self._conv_finish(conv)

    def conv_velx(self, ux, uy):
        conv = self._conv_term(ux, self.velx, [1, 0])
        conv += self._conv_term(uy, self.velx, [0, 1])
        return self._conv_finish(conv)

    def conv_vely(self, ux, uy):
        conv = self._conv_term(ux, self.vely, [1, 0])
        conv += self._conv_term(uy, self.vely, [0, 1])
        return self._conv_finish(conv)

    # ------------------------------------------------------------------ forward equations (steady_adjoint_eq.rs:123-181)
    def solve_velx(self, ux, uy, dt):
        self.zero_rhs()
        self.rhs += self.velx.to_ortho()
        self.rhs -= self.pres.gradient([1, 0], self.scale) * dt
        self.rhs -= self.conv_velx(ux, uy) * dt
        self.velx.vhat = self.solver_hholtz[0].solve(self.rhs)

    def solve_vely(self, ux, uy, dt):
        self.zero_rhs()
        self.rhs += self.vely.to_ortho()
        self.rhs -= self.pres.gradient([0, 1], self.scale) * dt
        self.rhs += self.temp.to_ortho() * dt          # buoyancy without the lift (steady_adjoint_eq.rs:151)
        self.rhs -= self.conv_vely(ux, uy) * dt
        self.vely.vhat = self.solver_hholtz[1].solve(self.rhs)

    def solve_temp(self, ux, uy, dt):
        self.zero_rhs()
        self.rhs += self.temp.to_ortho()
        ka = self.params["ka"]
        self.rhs += self.tempbc.gradient([2, 0], self.scale) * dt * ka
        self.rhs += self.tempbc.gradient([0, 2], self.scale) * dt * ka
        self.rhs -= self.conv_temp(ux, uy) * dt
        self.temp.vhat = self.solver_hholtz[2].solve(self.rhs)

    # ------------------------------------------------------------------ pressure (steady_adjoint_eq.rs:183-232)
    def correct_velocity(self, c):
        dp_dx = self.pseu.gradient([1, 0], self.scale) * (-c)
        dp_dy = self.pseu.gradient([0, 1], self.scale) * (-c)
        self.velx.vhat = self.velx.vhat + self.velx.space.from_ortho(dp_dx)
        self.vely.vhat = self.vely.vhat + self.vely.space.from_ortho(dp_dy)

    def update_pres(self, div, dt):
        a = -1.0 * self.params["nu"]
        b = 1.0 / dt
        self.pres.vhat = self.pres.vhat + div * a + self.pseu.to_ortho() * b

    def update_pres_adj(self, _div):
        b = 1.0 / self.dt
        self.pres_adj.vhat = self.pres_adj.vhat + self.pseu.to_ortho() * b

    def solve_pres(self, f):
        self.pseu.vhat = self.solver_pres.solve(f)
        self.pseu.vhat[0, 0] = 0.0

    # ------------------------------------------------------------------ convection, adjoint (steady_adjoint_eq.rs:234-325)
    def conv_velx_adjoint(self, ux, uy, temp_adj):
        conv = self._conv_term(ux, self.velx_adj, [1, 0])
        conv += self._conv_term(uy, self.velx_adj, [0, 1])
        conv += self._conv_term(ux, self.velx_adj, [1, 0])
        conv += self._conv_term(uy, self.vely_adj, [1, 0])
        conv -= self._conv_term(temp_adj, self.temp, [1, 0])
        conv -= self._conv_term(temp_adj, self.tempbc, [1, 0])
        return self._conv_finish(conv)

    def conv_vely_adjoint(self, ux, uy, temp_adj):
        conv = self._conv_term(ux, self.vely_adj, [1, 0])
        conv += self._conv_term(uy, self.vely_adj, [0, 1])
        conv += self._conv_term(ux, self.velx_adj, [0, 1])
        conv += self._conv_term(uy, self.vely_adj, [0, 1])
        conv -= self._conv_term(temp_adj, self.temp, [0, 1])
        conv -= self._conv_term(temp_adj, self.tempbc, [0, 1])
        return self._conv_finish(conv)

    def conv_temp_adjoint(self, ux, uy):
        conv = self._conv_term(ux, self.temp_adj, [1, 0])
        conv += self._conv_term(uy, self.temp_adj, [0, 1])
        return self._conv_finish(conv)

    # ------------------------------------------------------------------ adjoint equations (steady_adjoint_eq.rs:327-438)
    def solve_velx_adj(self, ux, uy, temp_adj):
        self.zero_rhs()
        self.rhs += self.velx.to_ortho()
        self.rhs -= self.pres_adj.gradient([1, 0], self.scale) * self.dt
        self.rhs += self.conv_velx_adjoint(ux, uy, temp_adj) * self.dt
        nu = self.params["nu"]
        self.rhs += self.velx_adj.gradient([2, 0], self.scale) * self.dt * nu
        self.rhs += self.velx_adj.gradient([0, 2], self.scale) * self.dt * nu
        self.velx.vhat = self.velx.space.from_ortho(self.rhs)

    def solve_vely_adj(self, ux, uy, temp_adj):
        self.zero_rhs()
        self.rhs += self.vely.to_ortho()
        self.rhs -= self.pres_adj.gradient([0, 1], self.scale) * self.dt
        self.rhs += self.conv_vely_adjoint(ux, uy, temp_adj) * self.dt
        nu = self.params["nu"]
        self.rhs += self.vely_adj.gradient([2, 0], self.scale) * self.dt * nu
        self.rhs += self.vely_adj.gradient([0, 2], self.scale) * self.dt * nu
        self.vely.vhat = self.vely.space.from_ortho(self.rhs)

    def solve_temp_adj(self, ux, uy):
        self.zero_rhs()
        self.rhs += self.temp.to_ortho()
        self.rhs += self.conv_temp_adjoint(ux, uy) * self.dt
        self.rhs += self.vely_adj.to_ortho() * self.dt
        ka = self.params["ka"]
        self.rhs += self.temp_adj.gradient([2, 0], self.scale) * self.dt * ka
        self.rhs += self.temp_adj.gradient([0, 2], self.scale) * self.dt * ka
        self.temp.vhat = self.temp.space.from_ortho(self.rhs)

    # ------------------------------------------------------------------ time step (steady_adjoint.rs:541-608)
    def update(self):
        # *** forward step to calculate the residual ***
        dtn = DT_NAVIER
        ux = self.velx.space.backward(self.velx.vhat)
        uy = self.vely.space.backward(self.vely.vhat)
        velx_old = self.velx.to_ortho()
        vely_old = self.vely.to_ortho()
        temp_old = self.temp.to_ortho()
        self.solve_velx(ux, uy, dtn)
        self.solve_vely(ux, uy, dtn)
        div = self.div()
        self.solve_pres(div)
        self.correct_velocity(1.0)
        self.update_pres(div, dtn)
        self.solve_temp(ux, uy, dtn)
        res_velx = (self.velx.to_ortho() - velx_old) / dtn
        res_vely = (self.vely.to_ortho() - vely_old) / dtn
        res_temp = (self.temp.to_ortho() - temp_old) / dtn
        self.velx_adj.vhat = -1.0 * self.solver_norm[0].solve(res_velx)
        self.vely_adj.vhat = -1.0 * self.solver_norm[1].solve(res_vely)
        self.temp_adj.vhat = -1.0 * self.solver_norm[2].solve(res_temp)
        # *** adjoint step ***
        ux = self.velx.space.backward(self.velx.vhat)
        uy = self.vely.space.backward(self.vely.vhat)
        temp_adj = self.temp_adj.space.backward(self.temp_adj.vhat)
        self.solve_velx_adj(ux, uy, temp_adj)
        self.solve_vely_adj(ux, uy, temp_adj)
        div = self.div()
        self.solve_pres(div)
        self.correct_velocity(1.0)
        self.update_pres_adj(div)
        self.solve_temp_adj(ux, uy)
        self.time += self.dt

    def exit(self):                                  # steady_adjoint.rs:624-638
        if np.isnan(self.div_norm()):
            return True
        return sum(self.norm_residual()) / 3.0 < RES_TOL

    # ------------------------------------------------------------------ outputs
    FIELDS = ("velx", "vely", "temp", "pres", "velx_adj", "vely_adj", "temp_adj", "pres_adj", "pseu")

    def physical_fields(self, names=("velx", "vely", "temp", "pres")):
        out = {}
        for k in names:
            f = getattr(self, k)
            f.backward()
            out[k] = f.v.copy()
        return out

    def spectral_fields(self, names=FIELDS):
        return {k: getattr(self, k).vhat.copy() for k in names}
