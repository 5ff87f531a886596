"""``Navier2DLnse`` of rustpde (Navier-Stokes linearised about mean fields) -- CPU oracle (test infrastructure; see
oracle/__init__.py).

Follows, in this order:
  mean fields       ``src/navier_stokes_lnse/meanfield.rs:20-56`` ("rbc" default), ``237-259`` (read: physical arrays + lift)
  constructors      ``src/navier_stokes_lnse/lnse.rs:98-176`` (confined), ``196-253`` (periodic)
  equations         ``src/navier_stokes_lnse/lnse_eq.rs`` (whole file)
  update() / exit() ``lnse.rs:263-288, 305-313``
  adjoint equations ``lnse_adj_eq.rs:16-94, 217-294``; adjoint step, adjoint gradient ``lnse_adj_grad.rs:43-225``;
  finite-difference gradient ``lnse_fd_grad.rs:31-157``; energy / l2_norm ``functions.rs:11-58``;
  steepest descent ``opt_routines.rs:16-56``
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

    @classmethod
    def new_hc(cls, space):
        """``new_hc_confined / new_hc_periodic`` (meanfield.rs:52-86, 154-188): no mean flow; per x a parabola in y with its
        vertex (value 0, slope 0) at the top wall and the value -0.5 cos(2 pi (x - x0) / L) at the bottom wall; forward and
        backward transformed like the reference does."""
        m = cls(space)
        x, y = m.temp.x
        f_x = -0.5 * np.cos(2.0 * np.pi * (x - x[0]) / (x[-1] - x[0]))
        a = f_x / (y[0] - y[-1]) ** 2
        m.temp.v[:, :] = a[:, None] * ((y - y[-1]) ** 2)[None, :]
        m.temp.forward()
        m.temp.backward()
        return m

    def set_physical(self, name, v):
        f = getattr(self, name)
        f.v = np.array(v, dtype=np.float64, copy=True)
        f.forward()


class Navier2DLnse:
    def __init__(self, nx, ny, ra, pr, dt, aspect, bc, periodic, eig_mode="full"):
        if bc not in ("rbc", "hc"):              # lnse.rs:115-119, 202-206 / nonlin.rs:117-121, 208-212
            raise ValueError(f"Boundary condition type {bc!r} not recognized!")
        temp_y = B.cheb_dirichlet if bc == "rbc" else B.cheb_dirichlet_neumann
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
            self.temp = Field2(S(bx(nx), temp_y(ny)))
        else:
            self.field = Field2(S(B.chebyshev(nx), B.chebyshev(ny)))
            self.velx = Field2(S(B.cheb_dirichlet(nx), B.cheb_dirichlet(ny)))
            self.vely = Field2(S(B.cheb_dirichlet(nx), B.cheb_dirichlet(ny)))
            self.pres = Field2(S(B.chebyshev(nx), B.chebyshev(ny)))
            self.pseu = Field2(S(B.cheb_neumann(nx), B.cheb_neumann(ny)))
            self.temp = Field2(S(B.cheb_neumann(nx), temp_y(ny)))
        for f in (self.velx, self.vely, self.temp, self.pres):
            f.scale(scale)
        self.mean = MeanFields.new_rbc(self.field.space) if bc == "rbc" else MeanFields.new_hc(self.field.space)   # meanfield.rs:112-119
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

    update_direct = update                       # lnse_adj_grad.rs:43-68 is the same sequence

    def reset_time(self):
        self.time = 0.0

    def exit(self):
        return bool(np.isnan(self.div_norm()))

    def init_random(self, amp, seed):
        """Seeded stand-in for ``init_random`` (``lnse.rs``: uniform(-amp, amp), order temp, velx, vely)."""
        rng = np.random.default_rng(seed)
        for f in (self.temp, self.velx, self.vely):
            f.v = rng.uniform(-amp, amp, size=f.v.shape)
            f.forward()

    def set_field_physical(self, name, v):
        f = getattr(self, name)
        f.v = np.array(v, dtype=np.float64, copy=True)
        f.forward()

    # ------------------------------------------------------------------ lnse_adj_eq.rs
    def _conv_adjoint(self, f, deriv_mean, velx, vely, temp, with_mean_gradients=True):
        """conv_velx_adjoint / conv_vely_adjoint / conv_temp_adjoint (lnse_adj_eq.rs:16-94):
        + U d/dx f* + V d/dy f*  - u* d_j U - v* d_j V - T* d_j Tm   (j = the component's direction; the temperature: no
        mean-gradient terms)."""
        self.mean.velx.backward()
        self.mean.vely.backward()
        um, vm = self.mean.velx.v, self.mean.vely.v
        conv = self._conv_term(um, f, [1, 0])
        conv += self._conv_term(vm, f, [0, 1])
        if with_mean_gradients:
            conv -= self._conv_term(velx, self.mean.velx, deriv_mean)
            conv -= self._conv_term(vely, self.mean.vely, deriv_mean)
            conv -= self._conv_term(temp, self.mean.temp, deriv_mean)
        self.field.v = conv
        self.field.forward()
        vhat = self.field.vhat
        vhat[vhat.shape[0] * 2 // 3:, :] = 0
        vhat[:, vhat.shape[1] * 2 // 3:] = 0
        return vhat.copy()

    def solve_velx_adj(self, velx, vely, temp):   # lnse_adj_eq.rs:217-238
        self.zero_rhs()
        self.rhs += self.velx.to_ortho()
        self.rhs -= self.pres.gradient([1, 0], self.scale) * self.dt
        self.rhs += self._conv_adjoint(self.velx, [1, 0], velx, vely, temp) * self.dt
        self.velx.vhat = self.solver_hholtz[0].solve(self.rhs)

    def solve_vely_adj(self, velx, vely, temp):   # lnse_adj_eq.rs:241-262
        self.zero_rhs()
        self.rhs += self.vely.to_ortho()
        self.rhs -= self.pres.gradient([0, 1], self.scale) * self.dt
        self.rhs += self._conv_adjoint(self.vely, [0, 1], velx, vely, temp) * self.dt
        self.vely.vhat = self.solver_hholtz[1].solve(self.rhs)

    def solve_temp_adj(self, velx, vely, temp, vely_vhat):   # lnse_adj_eq.rs:269-294
        self.zero_rhs()
        self.rhs += self.temp.to_ortho()
        self.rhs += self._conv_adjoint(self.temp, None, velx, vely, temp, with_mean_gradients=False) * self.dt
        self.rhs += vely_vhat * self.dt
        self.temp.vhat = self.solver_hholtz[2].solve(self.rhs)

    def update_adjoint(self):                    # lnse_adj_grad.rs:71-99
        uyhat = self.vely.to_ortho()
        self.velx.backward()
        self.vely.backward()
        self.temp.backward()
        velx, vely, temp = self.velx.v.copy(), self.vely.v.copy(), self.temp.v.copy()
        self.solve_velx_adj(velx, vely, temp)
        self.solve_vely_adj(velx, vely, temp)
        div = self.div()
        self.solve_pres(div)
        self.correct_velocity(1.0)
        self.update_pres(div)
        self.solve_temp_adj(velx, vely, temp, uyhat)
        self.time += self.dt

    # ------------------------------------------------------------------ lnse_adj_grad.rs / lnse_fd_grad.rs / functions.rs
    def _exit_grad(self, max_time, timestep, max_timestep=10_000_000):   # lnse_adj_grad.rs:204-225
        if self.time + self.dt * 1e-4 >= max_time:
            return True
        if timestep >= max_timestep:
            return True
        return bool(np.isnan(self.div_norm()))

    def energy(self, beta1, beta2):               # functions.rs:11-28
        self.velx.backward()
        self.vely.backward()
        self.temp.backward()
        return l2_norm(self.velx.v, self.velx.v, self.vely.v, self.vely.v, self.temp.v, self.temp.v, beta1, beta2)

    SUPPRESS_FORWARD_INFO = True                 # lnse_adj_grad.rs:128 passes suppress_io = true, nonlin_adj_grad.rs:137 false

    def _snapshot_refresh(self):
        """What a snapshot written inside the gradient loops does to the state (lnse_io.rs:73-91, 44-47): on the output interval
        (OUTPUT_INTERVALL = 1) `write` runs `backward()` on the fields -- the physical arrays grad_adjoint returns."""
        if (self.time + self.dt / 2.0) % 1.0 < self.dt:
            for f in (self.velx, self.vely, self.temp, self.pres):
                f.backward()
            return True
        return False

    def averages_of_squares(self):
        """u2, v2, t2 of the info line (lnse_io.rs:96-101) on the CURRENT state (the reference squares the physical arrays the state
        holds, which lag one step inside the adjoint loop)."""
        out = []
        for f in (self.velx, self.vely, self.temp):
            self.field.v = f.space.backward(f.vhat) ** 2
            out.append(self.field.average())
        return out

    def grad_adjoint(self, max_time, beta1, beta2, target=None, save_intervall=None):
        """lnse_adj_grad.rs:105-202 without the file output: forward loop, energy, the adjoint initial condition
        beta x (state - target), adjoint loop, gradient = -(the PHYSICAL arrays the state holds at the end -- the
        `backward()` of the adjoint step runs at the START of a step, so they are the adjoint fields one step before the end,
        :185-191 as written).  Returns (fun_val, (grad_u, grad_v, grad_t)) as physical arrays."""
        timestep = 0
        self.snapshots = []
        while True:
            self.update_direct()
            timestep += 1
            if save_intervall is not None:
                r = self.time % save_intervall
                if r < self.dt / 2.0 or r > save_intervall - self.dt / 2.0:
                    if self._snapshot_refresh():
                        self.snapshots.append(("flow", self.time))
            if self._exit_grad(max_time, timestep):
                break
        self.velx.backward()
        self.vely.backward()
        self.temp.backward()
        if target is None:
            u, v, t = self.velx.v, self.vely.v, self.temp.v
        else:
            u, v, t = self.velx.v - target.velx.v, self.vely.v - target.vely.v, self.temp.v - target.temp.v
        fun_val = l2_norm(u, u, v, v, t, t, beta1, beta2)
        if target is not None:
            self.velx.vhat = self.velx.vhat - self.velx.space.from_ortho(target.velx.vhat)
            self.vely.vhat = self.vely.vhat - self.vely.space.from_ortho(target.vely.vhat)
            self.temp.vhat = self.temp.vhat - self.temp.space.from_ortho(target.temp.vhat)
        self.velx.vhat = self.velx.vhat * beta1
        self.vely.vhat = self.vely.vhat * beta1
        self.temp.vhat = self.temp.vhat * beta2
        self.time = 0.0
        while True:
            self.update_adjoint()
            timestep += 1        # the counter is NOT reset between the loops (:117, :169)
            if save_intervall is not None and (self.time + self.dt / 2.0) % save_intervall < self.dt:
                if self._snapshot_refresh():
                    self.snapshots.append(("adjoint", self.time))
            if self._exit_grad(max_time, timestep):
                break
        fac = -1.0               # MAXIMIZE = false (:16)
        return fun_val, (fac * self.velx.v, fac * self.vely.v, fac * self.temp.v)

    def _integrate(self, max_time):               # src/lib.rs:187-219 without callbacks
        timestep = 0
        while True:
            self.update()
            timestep += 1
            if self.time + self.dt * 1e-4 >= max_time or timestep >= 10_000_000 or self.exit():
                break

    def grad_fd(self, max_time, beta1, beta2, points=None, eps=1e-5):
        """lnse_fd_grad.rs:31-157: one integration per perturbed grid point (`points`: an iterable of (field, i, j) to visit
        instead of every point -- a test device; unvisited entries stay 0)."""
        base_v = {k: getattr(self, k).v.copy() for k in ("velx", "vely", "temp")}
        base_h = {k: getattr(self, k).vhat.copy() for k in ("velx", "vely", "temp")}

        def reset():
            self.time = 0.0
            for k in base_v:
                getattr(self, k).v = base_v[k].copy()
                getattr(self, k).vhat = base_h[k].copy()
            for f in (self.pres, self.pseu):
                f.v = f.v * 0.0
                f.vhat = f.vhat * 0.0

        reset()
        self._integrate(max_time)
        e_base = self.energy(beta1, beta2)
        grads = {k: np.zeros_like(base_v[k]) for k in base_v}
        if points is None:
            points = [(k, i, j) for k in ("velx", "vely", "temp") for i in range(base_v[k].shape[0]) for j in range(base_v[k].shape[1])]
        for k, i, j in points:
            reset()
            f = getattr(self, k)
            f.v[i, j] += eps
            f.forward()
            self._integrate(max_time)
            grads[k][i, j] = 1.0 / eps * (self.energy(beta1, beta2) - e_base)
        return grads["velx"], grads["vely"], grads["temp"]

    def spectral_fields(self, names=("velx", "vely", "temp", "pres", "pseu")):
        return {k: getattr(self, k).vhat.copy() for k in names}


class _History:
    """One entry of ``field_history`` (nonlin_adj_grad.rs:69-77): clones of velx, vely, temp after their ``backward()``."""

    def __init__(self, nav):
        self.vhat = {k: getattr(nav, k).vhat.copy() for k in ("velx", "vely", "temp")}
        self.v = {k: getattr(nav, k).v.copy() for k in ("velx", "vely", "temp")}


class Navier2DNonLin(Navier2DLnse):
    """``Navier2DNonLin`` (src/navier_stokes_lnse/nonlin.rs, nonlin_eq.rs, nonlin_adj_eq.rs, nonlin_adj_grad.rs): the full
    non-linear equations for the deviation from the mean fields -- the LNSE step plus u . grad(u) + U . grad(U), the mean's
    diffusion and the mean temperature in the buoyancy -- with the history of the forward states that the adjoint loop's
    convection terms read back."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.field_history = []

    def _conv(self, ux, uy, mean_f, f):          # nonlin_eq.rs:59-134: eight terms
        self.mean.velx.backward()
        self.mean.vely.backward()
        um, vm = self.mean.velx.v, self.mean.vely.v
        conv = self._conv_term(ux, mean_f, [1, 0])
        conv += self._conv_term(uy, mean_f, [0, 1])
        conv += self._conv_term(um, f, [1, 0])
        conv += self._conv_term(vm, f, [0, 1])
        conv += self._conv_term(ux, f, [1, 0])
        conv += self._conv_term(uy, f, [0, 1])
        conv += self._conv_term(um, mean_f, [1, 0])
        conv += self._conv_term(vm, mean_f, [0, 1])
        self.field.v = conv
        self.field.forward()
        vhat = self.field.vhat
        vhat[vhat.shape[0] * 2 // 3:, :] = 0
        vhat[:, vhat.shape[1] * 2 // 3:] = 0
        return vhat.copy()

    def _mean_diffusion(self, mean_f, kappa):    # nonlin_eq.rs:204-206, 221-223, 236-238
        return (mean_f.gradient([2, 0], self.scale) + mean_f.gradient([0, 2], self.scale)) * (self.dt * kappa)

    def solve_velx(self, ux, uy):
        self.zero_rhs()
        self.rhs += self.velx.to_ortho()
        self.rhs -= self.pres.gradient([1, 0], self.scale) * self.dt
        self.rhs -= self._conv(ux, uy, self.mean.velx, self.velx) * self.dt
        self.rhs += self._mean_diffusion(self.mean.velx, self.params["nu"])
        self.velx.vhat = self.solver_hholtz[0].solve(self.rhs)

    def solve_vely(self, ux, uy, buoy):
        self.zero_rhs()
        self.rhs += self.vely.to_ortho()
        self.rhs -= self.pres.gradient([0, 1], self.scale) * self.dt
        self.rhs += buoy * self.dt
        self.rhs -= self._conv(ux, uy, self.mean.vely, self.vely) * self.dt
        self.rhs += self._mean_diffusion(self.mean.vely, self.params["nu"])
        self.vely.vhat = self.solver_hholtz[1].solve(self.rhs)

    def solve_temp(self, ux, uy):
        self.zero_rhs()
        self.rhs += self.temp.to_ortho()
        self.rhs -= self._conv(ux, uy, self.mean.temp, self.temp) * self.dt
        self.rhs += self._mean_diffusion(self.mean.temp, self.params["ka"])
        self.temp.vhat = self.solver_hholtz[2].solve(self.rhs)

    def update(self):                            # nonlin.rs:264-296
        that = self.temp.to_ortho() + self.mean.temp.to_ortho()
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

    def update_direct(self):                     # nonlin_adj_grad.rs:43-81: update() + the history entry
        self.update()
        self.velx.backward()
        self.vely.backward()
        self.temp.backward()
        self.field_history.append(_History(self))

    def _conv_adjoint_nl(self, f, deriv, velx, vely, temp, nl, with_gradients=True):   # nonlin_adj_eq.rs:16-118
        self.mean.velx.backward()
        self.mean.vely.backward()
        um, vm = self.mean.velx.v, self.mean.vely.v
        conv = self._conv_term(um, f, [1, 0])
        conv += self._conv_term(vm, f, [0, 1])
        if with_gradients:
            conv -= self._conv_term(velx, self.mean.velx, deriv)
            conv -= self._conv_term(vely, self.mean.vely, deriv)
            conv -= self._conv_term(temp, self.mean.temp, deriv)
        conv += self._conv_term(nl.v["velx"], f, [1, 0])
        conv += self._conv_term(nl.v["vely"], f, [0, 1])
        if with_gradients:
            for u, k in ((velx, "velx"), (vely, "vely"), (temp, "temp")):
                fld = getattr(self, k)
                conv -= u * self.field.space.backward(fld.space.gradient(nl.vhat[k], deriv, self.scale))
        self.field.v = conv
        self.field.forward()
        vhat = self.field.vhat
        vhat[vhat.shape[0] * 2 // 3:, :] = 0
        vhat[:, vhat.shape[1] * 2 // 3:] = 0
        return vhat.copy()

    SUPPRESS_FORWARD_INFO = False

    # nonlin_io.rs:145-198 with functions.rs:60-143: the diagnostics of the TOTAL fields (state + mean)
    def _total(self, name):
        return getattr(self, name).to_ortho() + getattr(self.mean, name).vhat

    def eval_nu(self):
        fld = self.field
        fld.vhat = self._total("temp")
        fld.vhat = fld.gradient([0, 1], None) * (-2.0 / self.scale[1])
        fld.backward()
        x_avg = fld.average_axis(0)
        return float((x_avg[-1] + x_avg[0]) / 2.0)

    def eval_nuvol(self):
        fld = self.field
        fld.vhat = self._total("temp")
        fld.backward()
        temp = fld.v.copy()
        fld.vhat = fld.gradient([0, 1], None) / (self.scale[1] * -1.0)
        fld.backward()
        dtdz = fld.v.copy()
        fld.vhat = self._total("vely")
        fld.backward()
        fld.v = (dtdz + fld.v * temp / self.params["ka"]) * 2.0 * self.scale[1]
        return fld.average()

    def eval_re(self):
        fld = self.field
        u = fld.space.backward(self._total("velx"))
        v = fld.space.backward(self._total("vely"))
        fld.v = np.sqrt(u ** 2 + v ** 2) * (2.0 * self.scale[1] / self.params["nu"])
        return fld.average()

    def update_adjoint(self, field_from_fwd=None):   # nonlin_adj_grad.rs:84-118 (None: the last history entry, like :190-193)
        nl = self.field_history.pop() if field_from_fwd is None else field_from_fwd
        uyhat = self.vely.to_ortho()
        self.velx.backward()
        self.vely.backward()
        self.temp.backward()
        velx, vely, temp = self.velx.v.copy(), self.vely.v.copy(), self.temp.v.copy()
        # solve_velx_adj / solve_vely_adj (nonlin_adj_eq.rs:127-165)
        self.zero_rhs()
        self.rhs += self.velx.to_ortho()
        self.rhs -= self.pres.gradient([1, 0], self.scale) * self.dt
        self.rhs += self._conv_adjoint_nl(self.velx, [1, 0], velx, vely, temp, nl) * self.dt
        self.velx.vhat = self.solver_hholtz[0].solve(self.rhs)
        self.zero_rhs()
        self.rhs += self.vely.to_ortho()
        self.rhs -= self.pres.gradient([0, 1], self.scale) * self.dt
        self.rhs += self._conv_adjoint_nl(self.vely, [0, 1], velx, vely, temp, nl) * self.dt
        self.vely.vhat = self.solver_hholtz[1].solve(self.rhs)
        div = self.div()
        self.solve_pres(div)
        self.correct_velocity(1.0)
        self.update_pres(div)
        # solve_temp_adj (nonlin_adj_eq.rs:168-188)
        self.zero_rhs()
        self.rhs += self.temp.to_ortho()
        self.rhs += self._conv_adjoint_nl(self.temp, None, velx, vely, temp, nl, with_gradients=False) * self.dt
        self.rhs += uyhat * self.dt
        self.temp.vhat = self.solver_hholtz[2].solve(self.rhs)
        self.time += self.dt


def l2_norm(a1, a2, b1, b2, c1, c2, beta1, beta2):
    """functions.rs:30-58: 0.5 * sum(beta1 a1 a2 + beta1 b1 b2 + beta2 c1 c2) -- a plain sum over the grid points."""
    return 0.5 * float((beta1 * a1 * a2 + beta1 * b1 * b2 + beta2 * c1 * c2).sum())


def steepest_descent_energy_constrained(velx_0, vely_0, temp_0, grad_velx, grad_vely, grad_temp, beta1, beta2, alpha):
    """opt_routines.rs:16-56: project the gradient perpendicular to the state, rotate the state by alpha towards it at constant
    energy.  Returns (velx_new, vely_new, temp_new) and the projected gradients (the reference modifies them in place)."""
    assert alpha <= 2.0 * np.pi, "alpha must be less than 2 pi"
    n = float(velx_0.size)
    e0 = l2_norm(velx_0, velx_0, vely_0, vely_0, temp_0, temp_0, beta1, beta2) / n
    eg = l2_norm(grad_velx, velx_0, grad_vely, vely_0, grad_temp, temp_0, beta1, beta2) / n
    ee = eg / e0
    gu, gv, gt = grad_velx - ee * velx_0, grad_vely - ee * vely_0, grad_temp - ee * temp_0
    eg = l2_norm(gu, gu, gv, gv, gt, gt, beta1, beta2) / n
    ee2 = np.sqrt(e0 / eg)
    ca, sa = np.cos(alpha), np.sin(alpha)
    return (velx_0 * ca + gu * (ee2 * sa), vely_0 * ca + gv * (ee2 * sa), temp_0 * ca + gt * (ee2 * sa)), (gu, gv, gt)
