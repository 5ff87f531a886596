"""rustpde_mpi_amd -- host-side mirror of rustpde's Navier2D interface over the HIP engine.

The reference is a compiled Rust crate; its Rust toolchain is not available in this image, so
the host side above the C ABI (include/rustpde_hip.h) is this thin Python mirror (used by the
tests and bench.py) plus the C++ engine inside the library.  Names, argument meaning and error
behaviour follow the reference:

    Navier2D.new_confined(nx, ny, ra, pr, dt, aspect, bc)    src/navier_stokes/navier.rs:215-308
    Navier2D.new_periodic(nx, ny, ra, pr, dt, aspect, bc)    src/navier_stokes/navier.rs:336-428
    .update() .get_time() .get_dt() .exit()                  trait Integrate, src/lib.rs:167-178
    integrate(pde, max_time, save_intervall)                 src/lib.rs:187-219
    .set_velocity .set_temperature .init_random .reset_time  src/navier_stokes/navier.rs:161-187
    .velx .vely .temp .pres .pseu (each with .v / .vhat / .x)  navier.rs:52-62, field.rs:59-72
    Space2 / HholtzAdi / Poisson                             funspace Space2, src/solver/*.rs

Where the reference panics (unknown bc, shape mismatch) this raises RpdeError.  The HIP library
is mandatory: importing works everywhere, but the first use raises if
rustpde_mpi_amd/librustpde_hip.so has not been built -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

from ._capi import Lib, RpdeError, as_f64, ptr

__all__ = ["Navier2D", "Navier2DMpi", "Navier2DAdjoint", "Navier2DLnse", "Space2", "HholtzAdi", "Poisson", "Hholtz", "integrate", "lib", "RpdeError",
           "chebyshev", "cheb_dirichlet", "cheb_neumann", "cheb_dirichlet_neumann", "fourier_r2c", "LIB_PATH", "Statistics"]

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "librustpde_hip.so")
_lib = None


def lib() -> Lib:
    """The in-tree HIP library (loaded on first use; raises RpdeError if it is missing)."""
    global _lib
    if _lib is None:
        _lib = Lib(LIB_PATH)
        if not _lib.is_device_build:
            raise RpdeError(f"{LIB_PATH} is not a HIP build")
    return _lib


PHYSICAL, SPECTRAL = 0, 1
CHEBYSHEV, CHEB_DIRICHLET, CHEB_NEUMANN, FOURIER_R2C, CHEB_DIRICHLET_NEUMANN = 0, 1, 2, 3, 4


def chebyshev(n): return (CHEBYSHEV, n)
def cheb_dirichlet(n): return (CHEB_DIRICHLET, n)
def cheb_neumann(n): return (CHEB_NEUMANN, n)
def fourier_r2c(n): return (FOURIER_R2C, n)
def cheb_dirichlet_neumann(n): return (CHEB_DIRICHLET_NEUMANN, n)   # axis 1 only (the "hc" temperature, navier.rs:245-248)


class _FieldView:
    """`Field2` members of the reference's Navier2D: `.v` (physical), `.vhat` (spectral), `.x`."""

    def __init__(self, nav: "Navier2D", name: str):
        self._nav, self._name = nav, name

    def _shape(self):
        r, c, z = C.c_int(), C.c_int(), C.c_int()
        self._nav._lib.call(self._nav._prefix + "_spectral_shape", self._nav._h, self._name.encode(),
                            C.byref(r), C.byref(c), C.byref(z))
        return r.value, c.value, bool(z.value)

    @property
    def v(self):
        out = np.empty((self._nav.nx, self._nav.ny))
        self._nav._lib.call(self._nav._prefix + "_get_field", self._nav._h, self._name.encode(), PHYSICAL,
                            ptr(out), out.size)
        return out

    @v.setter
    def v(self, value):
        a = as_f64(value)
        if a.shape != (self._nav.nx, self._nav.ny):
            raise RpdeError(f"physical field must have shape {(self._nav.nx, self._nav.ny)}")
        self._nav._lib.call(self._nav._prefix + "_set_field", self._nav._h, self._name.encode(), PHYSICAL,
                            ptr(a), a.size)

    @property
    def vhat(self):
        r, c, z = self._shape()
        out = np.empty((r, c * (2 if z else 1)))
        self._nav._lib.call(self._nav._prefix + "_get_field", self._nav._h, self._name.encode(), SPECTRAL,
                            ptr(out), out.size)
        return out.view(np.complex128) if z else out

    @vhat.setter
    def vhat(self, value):
        r, c, z = self._shape()
        a = as_f64(np.asarray(value, dtype=np.complex128 if z else np.float64))
        if a.shape != (r, c * (2 if z else 1)):
            raise RpdeError(f"spectral field {self._name} must have shape {(r, c)}")
        self._nav._lib.call(self._nav._prefix + "_set_field", self._nav._h, self._name.encode(), SPECTRAL,
                            ptr(a), a.size)

    @property
    def x(self):
        out = []
        for axis, n in enumerate((self._nav.nx, self._nav.ny)):
            g = np.empty(n)
            self._nav._lib.call("rpde_navier2d_get_grid", self._nav._h, axis, ptr(g), g.size)
            out.append(g)
        return out


class Navier2D:
    """Device-resident `Navier2D` (2-D Rayleigh-Benard convection, f64)."""
    _prefix = "rpde_navier2d"      # C-ABI family of this handle (the field views call <prefix>_get_field ...)

    def __init__(self, handle, nx, ny, periodic, library):
        self._h, self.nx, self.ny, self.periodic, self._lib = handle, nx, ny, periodic, library
        for name in ("velx", "vely", "temp", "pres", "pseu"):
            setattr(self, name, _FieldView(self, name))

    @classmethod
    def _new(cls, fn, nx, ny, ra, pr, dt, aspect, bc, device, library, periodic, comm=None, x_spectrum=None):
        library = library or lib()
        h = C.c_void_p()
        if x_spectrum is not None:
            # the Poisson solver's x eigenvalues supplied by the host: eigenbasis without LAPACK, bit-reproducible
            if periodic or (comm is not None and comm.size > 1):
                raise RpdeError("x_spectrum: confined, one device")
            lam = as_f64(np.asarray(x_spectrum, dtype=np.float64))
            library.call("rpde_navier2d_create_confined_with_spectrum", int(nx), int(ny), float(ra), float(pr), float(dt),
                         float(aspect), str(bc).encode(), int(device), ptr(lam), lam.size, C.byref(h))
        elif comm is not None and comm.size > 1 and getattr(comm, "native_rccl", False):
            # pencil-sharded engine, native transport: grouped ncclSend/ncclRecv on the engine's stream
            library.call("rpde_navier2d_create_sharded_rccl", int(periodic), int(nx), int(ny), float(ra),
                         float(pr), float(dt), float(aspect), str(bc).encode(), int(device),
                         int(comm.rank), int(comm.size), comm.unique_id(library), C.byref(h))
        elif comm is not None and comm.size > 1:
            # pencil-sharded engine (the reference's Navier2DMpi): `comm` supplies rank, size and the
            # all-to-all (rustpde_mpi_amd.dist.TorchComm); every rank passes the same arguments
            library.call("rpde_navier2d_create_sharded", int(periodic), int(nx), int(ny), float(ra),
                         float(pr), float(dt), float(aspect), str(bc).encode(), int(device),
                         int(comm.rank), int(comm.size), C.cast(comm.c_callback, C.c_void_p), None,
                         C.byref(h))
        else:
            library.call(fn, int(nx), int(ny), float(ra), float(pr), float(dt), float(aspect),
                         str(bc).encode(), int(device), C.byref(h))
        obj = cls(h, int(nx), int(ny), periodic, library)
        obj._comm = comm   # keeps the ctypes callback alive
        return obj

    # The reference's constructors end with `navier.init_random(0.1)` (navier.rs:305, 425; an unseeded
    # RNG).  The C ABI creates the engine with an all-zero state (a host calls rpde_navier2d_init_random
    # itself); this mirror does what the reference does, with a fixed seed so that runs are reproducible.
    # Pass init_random=None to keep the zero state.
    @classmethod
    def new_confined(cls, nx, ny, ra, pr, dt, aspect, bc, device=0, library=None, comm=None, init_random=0.1, seed=0,
                     x_spectrum=None):
        obj = cls._new("rpde_navier2d_create_confined", nx, ny, ra, pr, dt, aspect, bc, device,
                       library, False, comm, x_spectrum)
        if init_random is not None:
            obj.init_random(init_random, seed)
        return obj

    @classmethod
    def new_periodic(cls, nx, ny, ra, pr, dt, aspect, bc, device=0, library=None, comm=None, init_random=0.1, seed=0):
        obj = cls._new("rpde_navier2d_create_periodic", nx, ny, ra, pr, dt, aspect, bc, device,
                       library, True, comm)
        if init_random is not None:
            obj.init_random(init_random, seed)
        return obj

    def comm_stats(self):
        """(bytes this rank sends per step, exchanges per step) of the pencil all-to-alls."""
        b, n = C.c_double(), C.c_int()
        self._lib.call("rpde_navier2d_comm_stats", self._h, C.byref(b), C.byref(n))
        return b.value, n.value

    def __del__(self):
        try:
            if self._h:
                self._lib.call("rpde_navier2d_destroy", self._h)
                self._h = None
        except Exception:
            pass

    # ---- initial conditions
    def set_velocity(self, amp, m, n):
        self._lib.call("rpde_navier2d_set_velocity", self._h, float(amp), float(m), float(n))

    def set_temperature(self, amp, m, n):
        self._lib.call("rpde_navier2d_set_temperature", self._h, float(amp), float(m), float(n))

    def init_random(self, amp, seed=0):
        self._lib.call("rpde_navier2d_init_random", self._h, float(amp), int(seed))

    def reset_time(self):
        self._lib.call("rpde_navier2d_reset_time", self._h)

    # ---- trait Integrate
    def update(self, nsteps: int = 1):
        self._lib.call("rpde_navier2d_update", self._h, int(nsteps))

    def get_time(self):
        t = C.c_double()
        self._lib.call("rpde_navier2d_time", self._h, C.byref(t))
        return t.value

    def get_dt(self):
        t = C.c_double()
        self._lib.call("rpde_navier2d_dt", self._h, C.byref(t))
        return t.value

    def exit(self):
        f = C.c_int()
        self._lib.call("rpde_navier2d_exit", self._h, C.byref(f))
        return bool(f.value)

    def callback(self):
        """`Integrate::callback` (navier.rs:476-480): snapshot data/flow{time:0>8.2}.h5 (on
        `write_intervall` when set) + the diagnostics line on stdout and in data/info.txt."""
        self._lib.call("rpde_navier2d_callback", self._h)

    def callback_from_filename(self, flow_name, info_name, suppress_io=False, write_flow_intervall=None):
        """navier_io.rs:84-149."""
        self._lib.call("rpde_navier2d_callback_from_filename", self._h, str(flow_name).encode(), str(info_name).encode(),
                       int(bool(suppress_io)), -1.0 if write_flow_intervall is None else float(write_flow_intervall))

    @property
    def write_intervall(self):
        return getattr(self, "_write_intervall", None)

    @write_intervall.setter
    def write_intervall(self, value):
        self._write_intervall = value
        self._lib.call("rpde_navier2d_set_write_intervall", self._h, -1.0 if value is None else float(value))

    def write(self, filename):
        """`Navier2D::write` (navier_io.rs:44-62): HDF5 snapshot in the reference's layout."""
        self._lib.call("rpde_navier2d_write", self._h, str(filename).encode())

    def read(self, filename):
        """`Navier2D::read` (navier_io.rs:21-29): restore ux, uy, temp, pres and the time."""
        self._lib.call("rpde_navier2d_read", self._h, str(filename).encode())

    def write_unwrap(self, filename):
        try:
            self.write(filename)
        except RpdeError as exc:
            print(f"Error while writing file {filename!r}. Error: {exc}")

    def read_unwrap(self, filename):
        try:
            self.read(filename)
            print(f"Reading file {filename!r} was successfull.")
        except RpdeError as exc:
            print(f"Error while reading file {filename!r}. Error: {exc}")

    # ---- statistics (`pub statistics: Option<Statistics<T, S>>`, navier.rs:88)
    @property
    def statistics(self):
        return getattr(self, "_statistics", None)

    @statistics.setter
    def statistics(self, value):
        if value is not None and value._nav is not self:
            raise RpdeError("Statistics belong to the Navier2D they were created from")
        # the engine's callback acts on `Some(statistics)` only (navier_io.rs:105); None detaches, the data stays
        if value is not None or getattr(self, "_statistics", None) is not None:
            self._lib.call("rpde_navier2d_statistics_attach", self._h, 1 if value is not None else 0)
        self._statistics = value

    # ---- extras
    def div_norm(self):
        v = C.c_double()
        self._lib.call("rpde_navier2d_div_norm", self._h, C.byref(v))
        return v.value

    def diagnostics(self):
        """(Nu, Nuvol, Re) as eval_nu / eval_nuvol / eval_re of the reference."""
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        self._lib.call("rpde_navier2d_diagnostics", self._h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def eval_nu(self): return self.diagnostics()[0]
    def eval_nuvol(self): return self.diagnostics()[1]
    def eval_re(self): return self.diagnostics()[2]

    def last_update_ms(self):
        v = C.c_double()
        self._lib.call("rpde_navier2d_last_update_ms", self._h, C.byref(v))
        return v.value

    def profile(self, nsteps=1):
        """Per-launch HIP-event profile: list of dicts (tag, launches, ms_total, bytes, flops)."""
        buf = C.create_string_buffer(1 << 16)
        self._lib.call("rpde_navier2d_profile", self._h, int(nsteps), buf, len(buf))
        rows = []
        for line in buf.value.decode().splitlines():
            tag, n, ms, by, fl = line.split("\t")
            rows.append({"tag": tag, "launches": int(n), "ms_total": float(ms), "bytes": float(by),
                         "flops": float(fl)})
        return rows

    def schedule(self):
        """The launches of one step in issue order: list of (tag, algorithmic bytes, flops, kernel
        dispatches behind the launch -- a column scan is 5 or 3 kernels --, kind of kernel: "line program",
        "whole-line transform" ... -- which form of a stage this engine chose, see RPDE_S1_LINE / RPDE_DCT_LINE)."""
        buf = C.create_string_buffer(1 << 16)
        self._lib.call("rpde_navier2d_describe_step", self._h, buf, len(buf))
        return [(t, float(b), float(f), int(n), kind) for t, b, f, n, kind in
                (line.split("\t") for line in buf.value.decode().splitlines())]

    def trace_launch(self, tag: str):
        """Diagnostics: one step with the first line program whose tag contains `tag` instrumented.
        Returns (tag, workgroups, span_ms, rows); rows = (id, name, mean, p10, median, p90): shader clocks
        since the previous mark; id >= 0: op id starts, -1: a barrier inside the op, -2 (first row): the
        whole program."""
        buf = C.create_string_buffer(1 << 17)
        self._lib.call("rpde_navier2d_trace_launch", self._h, tag.encode(), buf, len(buf))
        lines = buf.value.decode().splitlines()
        t, n, span, _ = lines[0].split("\t")
        rows = [(int(a), b, float(c), float(d), float(e), float(f)) for a, b, c, d, e, f in (l.split("\t") for l in lines[1:])]
        return t, int(n), float(span), rows

    def set_timed_tag(self, tag: str):
        self._lib.call("rpde_navier2d_set_timed_tag", self._h, tag.encode())

    def get_timed(self):
        ms, n = C.c_double(), C.c_long()
        self._lib.call("rpde_navier2d_get_timed", self._h, C.byref(ms), C.byref(n))
        return ms.value, n.value

    @property
    def params(self):
        out = {}
        for k in ("ra", "pr", "nu", "ka"):
            v = C.c_double()
            self._lib.call("rpde_navier2d_param", self._h, k.encode(), C.byref(v))
            out[k] = v.value
        return out

    def physical_fields(self):
        return {k: getattr(self, k).v for k in ("velx", "vely", "temp", "pres")}

    def poisson_eigenbasis(self):
        """(lam, fwd, bwd) of the pressure solver's x eigen-decomposition as the reference's
        FdmaTensor holds it (fdma_tensor.rs:123-127); confined engines only."""
        m = self.nx - 2
        lam, fwd, bwd = np.empty(m), np.empty((m, m)), np.empty((m, m))
        self._lib.call("rpde_navier2d_poisson_eigenbasis", self._h, ptr(lam), ptr(fwd), ptr(bwd), m)
        return lam, fwd, bwd

    def integrate(self, max_time, exit_check_every=1):
        n = C.c_long()
        self._lib.call("rpde_navier2d_integrate", self._h, float(max_time), int(exit_check_every),
                       C.byref(n))
        return n.value


MAX_TIMESTEP = 10_000_000


def integrate(pde, max_time, save_intervall=None):
    """`rustpde::integrate` (src/lib.rs:187-219): update / callback / break tests."""
    timestep = 0
    eps_dt = pde.get_dt() * 1e-4
    while True:
        pde.update()
        timestep += 1
        if save_intervall is not None:
            # lib.rs:196-203 verbatim: both sides of a multiple of the save interval count
            t, dt = pde.get_time(), pde.get_dt()
            if (t % save_intervall) < dt / 2.0 or (t % save_intervall) > save_intervall - dt / 2.0:
                pde.callback()
        if pde.get_time() + eps_dt >= max_time:
            break
        if timestep >= MAX_TIMESTEP:
            break
        if pde.exit():
            break
    return timestep


# ------------------------------------------------------------------------------------------------
class Space2:
    """funspace `Space2::new(&base0, &base1)` on the device (operator-level C ABI)."""

    def __init__(self, base0, base1, device=0, library=None):
        self._lib = library or lib()
        self._h = C.c_void_p()
        self._lib.call("rpde_space2_create", base0[0], base0[1], base1[0], base1[1], int(device),
                       C.byref(self._h))

    def __del__(self):
        try:
            if self._h:
                self._lib.call("rpde_space2_destroy", self._h)
                self._h = None
        except Exception:
            pass

    def shape(self, which):
        r, c, z = C.c_int(), C.c_int(), C.c_int()
        self._lib.call("rpde_space2_shape", self._h, {"physical": 0, "spectral": 1, "ortho": 2}[which],
                       C.byref(r), C.byref(c), C.byref(z))
        return r.value, c.value, bool(z.value)

    def _out(self, which):
        r, c, z = self.shape(which)
        return np.empty((r, c * (2 if z else 1))), z

    def _run(self, fn, a, which_out, *extra):
        a = as_f64(a)
        out, z = self._out(which_out)
        self._lib.call(fn, self._h, ptr(a), a.size, *extra, ptr(out), out.size)
        return out.view(np.complex128) if z else out

    def forward(self, v): return self._run("rpde_space2_forward", v, "spectral")
    def backward(self, vhat): return self._run("rpde_space2_backward", vhat, "physical")
    def to_ortho(self, vhat): return self._run("rpde_space2_to_ortho", vhat, "ortho")
    def from_ortho(self, c): return self._run("rpde_space2_from_ortho", c, "spectral")

    def gradient(self, vhat, deriv, scale=None):
        s = scale or (1.0, 1.0)
        return self._run("rpde_space2_gradient", vhat, "ortho", int(deriv[0]), int(deriv[1]),
                         float(s[0]), float(s[1]))


class _Solver:
    _create = _solve = _destroy = ""

    def __init__(self, space: Space2, c):
        self._space, self._lib = space, space._lib
        self._h = C.c_void_p()
        self._lib.call(self._create, space._h, float(c[0]), float(c[1]), C.byref(self._h))

    def __del__(self):
        try:
            if self._h:
                self._lib.call(self._destroy, self._h)
                self._h = None
        except Exception:
            pass

    def solve(self, rhs_ortho):
        a = as_f64(rhs_ortho)
        out, z = self._space._out("spectral")
        self._lib.call(self._solve, self._h, ptr(a), a.size, ptr(out), out.size)
        return out.view(np.complex128) if z else out


class HholtzAdi(_Solver):
    """`HholtzAdi::new(&field, [c0, c1])` + `solve` (src/solver/hholtz_adi.rs:48-76,149-169)."""
    _create, _solve, _destroy = "rpde_hholtz_adi_create", "rpde_hholtz_adi_solve", "rpde_hholtz_adi_destroy"


class Poisson(_Solver):
    """`Poisson::new(&field, [c0, c1])` + `solve` (src/solver/poisson.rs:54-94,195-236)."""
    _create, _solve, _destroy = "rpde_poisson_create", "rpde_poisson_solve", "rpde_poisson_destroy"

    def __init__(self, space: Space2, c, x_spectrum=None):
        if x_spectrum is None:
            super().__init__(space, c)
            return
        self._space, self._lib = space, space._lib
        self._h = C.c_void_p()
        lam = as_f64(np.asarray(x_spectrum, dtype=np.float64))
        self._lib.call("rpde_poisson_create_with_spectrum", space._h, float(c[0]), float(c[1]), ptr(lam), lam.size, C.byref(self._h))

    def eigenbasis(self):
        """(lam, fwd, bwd) of the x eigen-decomposition (fdma_tensor.rs:123-127)."""
        m = self._space.shape("spectral")[0]
        lam, fwd, bwd = np.empty(m), np.empty((m, m)), np.empty((m, m))
        self._lib.call("rpde_poisson_eigenbasis", self._h, ptr(lam), ptr(fwd), ptr(bwd), m)
        return lam, fwd, bwd


def poisson_x_spectrum(base, c0, library=None):
    """The n - 2 x eigenvalues `Poisson::new` computes for a Chebyshev axis `base` = (kind, n) (host only: LAPACK dgeev,
    values only); order [even block | odd block], each descending."""
    library = library or lib()
    kind, n = base
    lam = np.empty(n - 2)
    library.call("rpde_poisson_x_spectrum", int(kind), int(n), float(c0), ptr(lam), lam.size)
    return lam


def poisson_x_eigenbasis_from_spectrum(base, c0, lam, library=None):
    """(refined lam, fwd, bwd) for given (approximate) eigenvalues: the bit-reproducible eigenbasis an engine created with `x_spectrum=lam` uses
    (host only, no LAPACK) -- the format of `Navier2D.poisson_eigenbasis()` / the oracle's `eig_override`."""
    library = library or lib()
    kind, n = base
    lam = as_f64(np.asarray(lam, dtype=np.float64))
    m = n - 2
    ref, fwd, bwd = np.empty(m), np.empty((m, m)), np.empty((m, m))
    library.call("rpde_poisson_x_eigenbasis_from_spectrum", int(kind), int(n), float(c0), ptr(lam), lam.size, ptr(ref), ptr(fwd), ptr(bwd))
    return ref, fwd, bwd


class Hholtz(_Solver):
    """`Hholtz::new(&field, [c0, c1])` + `solve`: (I - c D2) vhat = A f with the tensor solver (src/solver/hholtz.rs:72-106,164-187)."""
    _create, _solve, _destroy = "rpde_hholtz_create", "rpde_hholtz_solve", "rpde_hholtz_destroy"


class Navier2DAdjoint:
    """Device-resident `Navier2DAdjoint` (src/navier_stokes/steady_adjoint.rs): adjoint descent to steady states.
    Same spelling as the reference: `new_confined / new_periodic`, `set_velocity`, `set_temperature`, `update()`, `exit()`,
    `.velx.vhat` ... `.temp_adj.v`, `div_norm()`, `norm_residual()`; `integrate(pde, max_time, None)` drives it."""
    _prefix = "rpde_adjoint2d"
    FIELDS = ("velx", "vely", "temp", "pres", "pseu", "velx_adj", "vely_adj", "temp_adj", "pres_adj", "tempbc")

    def __init__(self, handle, nx, ny, periodic, library):
        self._h, self.nx, self.ny, self.periodic, self._lib = handle, nx, ny, periodic, library
        for name in self.FIELDS:
            setattr(self, name, _FieldView(self, name))

    @classmethod
    def _new(cls, fn, nx, ny, ra, pr, dt, aspect, bc, device, library, periodic):
        library = library or lib()
        h = C.c_void_p()
        library.call(fn, int(nx), int(ny), float(ra), float(pr), float(dt), float(aspect), str(bc).encode(), int(device), C.byref(h))
        return cls(h, nx, ny, periodic, library)

    @classmethod
    def new_confined(cls, nx, ny, ra, pr, dt, aspect, bc, device=0, library=None):
        return cls._new("rpde_adjoint2d_create_confined", nx, ny, ra, pr, dt, aspect, bc, device, library, False)

    @classmethod
    def new_periodic(cls, nx, ny, ra, pr, dt, aspect, bc, device=0, library=None):
        return cls._new("rpde_adjoint2d_create_periodic", nx, ny, ra, pr, dt, aspect, bc, device, library, True)

    def __del__(self):
        try:
            if self._h:
                self._lib.call(self._prefix + "_destroy", self._h)
                self._h = None
        except Exception:
            pass

    def set_velocity(self, amp, m, n):
        self._lib.call(self._prefix + "_set_velocity", self._h, float(amp), float(m), float(n))

    def set_temperature(self, amp, m, n):
        self._lib.call(self._prefix + "_set_temperature", self._h, float(amp), float(m), float(n))

    def reset_time(self):
        self._lib.call(self._prefix + "_reset_time", self._h)

    def update(self, nsteps: int = 1):
        self._lib.call(self._prefix + "_update", self._h, int(nsteps))

    def get_time(self):
        t = C.c_double()
        self._lib.call(self._prefix + "_time", self._h, C.byref(t))
        return t.value

    def get_dt(self):
        t = C.c_double()
        self._lib.call(self._prefix + "_dt", self._h, C.byref(t))
        return t.value

    def exit(self):
        f = C.c_int()
        self._lib.call(self._prefix + "_exit", self._h, C.byref(f))
        return bool(f.value)

    def write(self, filename):
        """`Navier2DAdjoint::write` (steady_adjoint_io.rs:48-71)."""
        self._lib.call(self._prefix + "_write", self._h, str(filename).encode())

    def read(self, filename):
        """`Navier2DAdjoint::read` (steady_adjoint_io.rs:22-33): ux, uy, temp, time -- also from a `Navier2D` snapshot."""
        self._lib.call(self._prefix + "_read", self._h, str(filename).encode())
        print(f" <== {str(filename)!r}")

    def write_unwrap(self, filename):
        try:
            self.write(filename)
        except RpdeError as e:
            print(f"Error while writing file {str(filename)!r}. Error: {e}", file=sys.stderr)

    def read_unwrap(self, filename):
        try:
            self.read(filename)
            print(f"Reading file {str(filename)!r} was successfull.")
        except RpdeError as e:
            print(f"Error while reading file {str(filename)!r}. Error: {e}", file=sys.stderr)

    write_intervall = None

    def callback(self):
        """`Integrate::callback` (steady_adjoint.rs:616-620) -> `callback_from_filename` (steady_adjoint_io.rs:84-143):
        data/adjoint{time:0>8.2}.h5 on `write_intervall`, |div| and the residual norms on stdout.  (The Nu / Nuv / Re columns
        of data/info_adjoint.txt need the diagnostics reductions of Navier2D; not part of this slice.)"""
        os.makedirs("data", exist_ok=True)
        t, dt = self.get_time(), self.get_dt()
        name = "data/adjoint{:0>8.2f}.h5".format(t)
        if self.write_intervall is None or (t + dt / 2.0) % self.write_intervall < dt:
            self.write_unwrap(name)
        ru, rv, rt = self.norm_residual()
        print("time = {:4.2f}      |div| = {:4.2e}".format(t, self.div_norm()))
        print("|U| = {:10.2e}\n|V| = {:10.2e}\n|T| = {:10.2e}".format(ru, rv, rt))

    def div_norm(self):
        d = C.c_double()
        self._lib.call(self._prefix + "_div_norm", self._h, C.byref(d))
        return d.value

    def norm_residual(self):
        r = (C.c_double * 3)()
        self._lib.call(self._prefix + "_norm_residual", self._h, r)
        return [r[0], r[1], r[2]]

    @property
    def params(self):
        out = {}
        for k in ("ra", "pr", "nu", "ka"):
            v = C.c_double()
            self._lib.call(self._prefix + "_param", self._h, k.encode(), C.byref(v))
            out[k] = v.value
        return out

    def physical_fields(self, names=("velx", "vely", "temp", "pres")):
        return {k: getattr(self, k).v for k in names}

    def spectral_fields(self, names=FIELDS[:9]):
        return {k: getattr(self, k).vhat for k in names}


class Navier2DLnse(Navier2DAdjoint):
    """Device-resident `Navier2DLnse` (src/navier_stokes_lnse/lnse.rs): the Navier-Stokes step linearised about mean fields.
    `new_confined / new_periodic(nx, ny, ra, pr, dt, aspect, bc, mean_file=None)`; `mean_file=None` looks for "mean.h5" like
    the reference and falls back to the boundary condition's default mean; `.mean_velx.v` etc. read / assign the mean fields."""
    _prefix = "rpde_lnse2d"
    FIELDS = ("velx", "vely", "temp", "pres", "pseu", "tempbc")

    class _Mean:
        def __init__(self, nav, name):
            self._nav, self._name = nav, name

        @property
        def v(self):
            out = np.empty((self._nav.nx, self._nav.ny))
            self._nav._lib.call("rpde_lnse2d_get_mean", self._nav._h, self._name.encode(), ptr(out), out.size)
            return out

        @v.setter
        def v(self, value):
            a = as_f64(value)
            if a.shape != (self._nav.nx, self._nav.ny):
                raise RpdeError(f"mean field must have shape {(self._nav.nx, self._nav.ny)}")
            self._nav._lib.call("rpde_lnse2d_set_mean", self._nav._h, self._name.encode(), ptr(a), a.size)

    def __init__(self, handle, nx, ny, periodic, library):
        super().__init__(handle, nx, ny, periodic, library)
        for name in ("velx", "vely", "temp"):
            setattr(self, "mean_" + name, Navier2DLnse._Mean(self, name))

    @classmethod
    def _new(cls, fn, nx, ny, ra, pr, dt, aspect, bc, device, library, periodic, mean_file=None):
        library = library or lib()
        h = C.c_void_p()
        library.call(fn, int(nx), int(ny), float(ra), float(pr), float(dt), float(aspect), str(bc).encode(),
                     None if mean_file is None else str(mean_file).encode(), int(device), C.byref(h))
        return cls(h, nx, ny, periodic, library)

    @classmethod
    def new_confined(cls, nx, ny, ra, pr, dt, aspect, bc, device=0, library=None, mean_file=None):
        return cls._new("rpde_lnse2d_create_confined", nx, ny, ra, pr, dt, aspect, bc, device, library, False, mean_file)

    @classmethod
    def new_periodic(cls, nx, ny, ra, pr, dt, aspect, bc, device=0, library=None, mean_file=None):
        return cls._new("rpde_lnse2d_create_periodic", nx, ny, ra, pr, dt, aspect, bc, device, library, True, mean_file)

    def norm_residual(self):
        raise RpdeError("Navier2DLnse has no residual (steady_adjoint.rs only)")

    def callback(self):
        """`Integrate::callback` (lnse.rs:298-303 / nonlin.rs:306-310)."""
        self.callback_from_filename("data/flow{:0>8.2f}.h5".format(self.get_time()), "data/info.txt", False, None)

    def callback_from_filename(self, flow_name, info_name, suppress_io=False, write_flow_intervall=None):
        """lnse_io.rs:73-126 / nonlin_io.rs:72-142: snapshot on the write interval (None: OUTPUT_INTERVALL = 1), the diagnostics
        line on stdout and in the info file."""
        self._lib.call("rpde_lnse2d_callback_from_filename", self._h, str(flow_name).encode(), str(info_name).encode(),
                       int(bool(suppress_io)), -1.0 if write_flow_intervall is None else float(write_flow_intervall))

    def diagnostics(self):
        """|div|, Nu, Nuv, Re (Navier2DNonLin: eval_nu / eval_nuvol / eval_re of state + mean, nonlin_io.rs:145-198; NaN for
        Navier2DLnse), and the weighted averages <u^2>, <v^2>, <T^2> of the info line."""
        out = np.empty(7)
        self._lib.call("rpde_lnse2d_diagnostics", self._h, ptr(out))
        return dict(zip(("div", "nu", "nuvol", "re", "u2", "v2", "t2"), out.tolist()))

    def spectral_fields(self, names=("velx", "vely", "temp", "pres", "pseu")):
        return {k: getattr(self, k).vhat for k in names}

    # ---- adjoint-based sensitivity of the final energy (lnse_adj_grad.rs, lnse_fd_grad.rs) ----
    def update_direct(self, nsteps: int = 1):
        """`Navier2DLnse::update_direct` (lnse_adj_grad.rs:43-68): the same sequence as `update`; `Navier2DNonLin`: + one entry
        of the field history per step (nonlin_adj_grad.rs:43-81)."""
        self._lib.call("rpde_lnse2d_update_direct", self._h, int(nsteps))

    def update_adjoint(self, nsteps: int = 1):
        """`Navier2DLnse::update_adjoint` (lnse_adj_grad.rs:71-99)."""
        self._lib.call("rpde_lnse2d_update_adjoint", self._h, int(nsteps))

    def integrate(self, max_time):
        n = C.c_long()
        self._lib.call("rpde_lnse2d_integrate", self._h, float(max_time), C.byref(n))
        return n.value

    def _target(self, target):
        if target is None:
            return [None, None, None]
        arrs = [as_f64(getattr(target, k).v if hasattr(target, k) else target[i]) for i, k in enumerate(("velx", "vely", "temp"))]
        for a in arrs:
            if a.shape != (self.nx, self.ny):
                raise RpdeError(f"target fields must have shape {(self.nx, self.ny)}")
        return arrs

    def energy(self, beta1, beta2, target=None):
        """`functions::energy` (functions.rs:11-28) of the state (minus the target's physical fields, lnse_adj_grad.rs:141-155)."""
        t = self._target(target)
        e = C.c_double()
        self._lib.call("rpde_lnse2d_energy", self._h, float(beta1), float(beta2), *[None if a is None else ptr(a) for a in t],
                       self.nx * self.ny, C.byref(e))
        return e.value

    def grad_adjoint(self, max_time, save_intervall, beta1, beta2, target=None, filename="data/grad_adjoint.h5"):
        """`Navier2DLnse::grad_adjoint` (lnse_adj_grad.rs:105-202) -> (fun_val, (grad_u, grad_v, grad_t)); the gradients are the
        physical arrays (`.v` of the reference's Field2s).  `target`: an object with `.velx.v / .vely.v / .temp.v` (MeanFields) or
        three arrays.  `filename`: the reference writes "data/grad_adjoint.h5" unconditionally; None skips the file.
        `save_intervall`: snapshots "data/flow*.h5" / "data/adjoint*.h5" and the info files during the two loops (directory
        "data" of the working directory, like the reference)."""
        t = self._target(target)
        n = self.nx * self.ny
        gu, gv, gt = (np.empty((self.nx, self.ny)) for _ in range(3))
        fun, steps = C.c_double(), C.c_long()
        if filename is not None and os.path.dirname(filename):
            os.makedirs(os.path.dirname(filename), exist_ok=True)
        self._lib.call("rpde_lnse2d_grad_adjoint", self._h, float(max_time), -1.0 if save_intervall is None else float(save_intervall),
                       float(beta1), float(beta2),
                       *[None if a is None else ptr(a) for a in t], n, None if filename is None else str(filename).encode(),
                       C.byref(fun), ptr(gu), ptr(gv), ptr(gt), C.byref(steps))
        return fun.value, (gu, gv, gt)

    def grad_fd(self, max_time, save_intervall, beta1, beta2, points=None, filename="data/grad_fd.h5"):
        """`Navier2DLnse::grad_fd` (lnse_fd_grad.rs:31-157): one integration per perturbed grid point (3 nx ny integrations --
        "should only be used for testing").  `points`: iterable of (field, i, j) with field in velx / vely / temp to visit
        instead of every point (the other entries stay 0).  `save_intervall`: None, or the interval on which the BASE run calls
        `callback()` (data/flow{time:0>8.2}.h5, data/info.txt in the working directory)."""
        pts, npts = None, 0
        if points is not None:
            code = {"velx": 0, "vely": 1, "temp": 2}
            arr = np.ascontiguousarray([(code[k] if isinstance(k, str) else int(k), int(i), int(j)) for k, i, j in points],
                                       dtype=np.intc).reshape(-1, 3)
            pts, npts = arr.ctypes.data_as(C.POINTER(C.c_int)), arr.shape[0]
        gu, gv, gt = (np.empty((self.nx, self.ny)) for _ in range(3))
        if filename is not None and os.path.dirname(filename):
            os.makedirs(os.path.dirname(filename), exist_ok=True)
        if save_intervall is not None:      # Some(dt): the base run writes its snapshot series (lnse_fd_grad.rs:54)
            os.makedirs("data", exist_ok=True)
        self._lib.call("rpde_lnse2d_grad_fd_save", self._h, float(max_time), -1.0 if save_intervall is None else float(save_intervall),
                       float(beta1), float(beta2), pts, npts, self.nx * self.ny,
                       None if filename is None else str(filename).encode(), ptr(gu), ptr(gv), ptr(gt))
        return gu, gv, gt


class Navier2DNonLin(Navier2DLnse):
    """Device-resident `Navier2DNonLin` (src/navier_stokes_lnse/nonlin.rs): the non-linear equations for the deviation from the
    mean fields, with the history of forward states (in HBM) that `update_adjoint` / `grad_adjoint` read back.  Same methods as
    `Navier2DLnse`; `update_adjoint()` consumes the last history entry (the reference passes it explicitly, taken from
    `field_history` the same way: nonlin_adj_grad.rs:190-193)."""

    @classmethod
    def new_confined(cls, nx, ny, ra, pr, dt, aspect, bc, device=0, library=None, mean_file=None):
        return cls._new("rpde_nonlin2d_create_confined", nx, ny, ra, pr, dt, aspect, bc, device, library, False, mean_file)

    @classmethod
    def new_periodic(cls, nx, ny, ra, pr, dt, aspect, bc, device=0, library=None, mean_file=None):
        return cls._new("rpde_nonlin2d_create_periodic", nx, ny, ra, pr, dt, aspect, bc, device, library, True, mean_file)

    @property
    def field_history_len(self):
        n = C.c_long()
        self._lib.call("rpde_lnse2d_history_len", self._h, C.byref(n))
        return n.value

    def clear_field_history(self):
        self._lib.call("rpde_lnse2d_clear_history", self._h)


def l2_norm(a1, a2, b1, b2, c1, c2, beta1, beta2, library=None):
    """`functions::l2_norm` (src/navier_stokes_lnse/functions.rs:30-58)."""
    library = library or lib()
    arrs = [as_f64(a) for a in (a1, a2, b1, b2, c1, c2)]
    if any(a.shape != arrs[0].shape for a in arrs):
        raise RpdeError("l2_norm: shapes differ")
    out = C.c_double()
    library.call("rpde_l2_norm", arrs[0].size, *[ptr(a) for a in arrs], float(beta1), float(beta2), C.byref(out))
    return out.value


def steepest_descent_energy_constrained(velx_0, vely_0, temp_0, grad_velx, grad_vely, grad_temp, velx_new, vely_new, temp_new,
                                        beta1, beta2, alpha, library=None):
    """`opt_routines::steepest_descent_energy_constrained` (opt_routines.rs:16-56): the reference's argument list -- the
    gradients (C-contiguous float64 arrays) are projected IN PLACE, the rotated state is written into `*_new`."""
    library = library or lib()
    ins = [as_f64(a) for a in (velx_0, vely_0, temp_0)]
    io = [grad_velx, grad_vely, grad_temp, velx_new, vely_new, temp_new]
    for a in io:
        if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous and a.shape == ins[0].shape):
            raise RpdeError("steepest_descent_energy_constrained: gradients and outputs must be C-contiguous float64 arrays of the state's shape")
    library.call("rpde_steepest_descent_energy_constrained", ins[0].size, *[ptr(a) for a in ins], *[ptr(a) for a in io],
                 float(beta1), float(beta2), float(alpha))


def transpose(a, device=0, library=None):
    library = library or lib()
    a = np.asarray(a)
    z = np.iscomplexobj(a)
    f = as_f64(a)
    rows, cols = a.shape
    out = np.empty((cols, rows * (2 if z else 1)))
    library.call("rpde_transpose", ptr(f), rows, cols, 2 if z else 1, ptr(out), int(device))
    return out.view(np.complex128) if z else out


def gemm(a, b, transb=False, device=0, library=None):
    library = library or lib()
    a, b = as_f64(a), as_f64(b)
    M, K = a.shape
    N = b.shape[0] if transb else b.shape[1]
    out = np.empty((M, N))
    library.call("rpde_gemm", M, N, K, ptr(a), ptr(b), 1 if transb else 0, ptr(out), int(device))
    return out


class h5:
    """The HDF5 subset of the snapshots through the library's own reader / writer (csrc/h5lite)."""

    @staticmethod
    def paths(filename, library=None):
        library = library or lib()
        buf = C.create_string_buffer(1 << 16)
        library.call("rpde_h5_list", str(filename).encode(), buf, len(buf))
        return buf.value.decode().split()

    @staticmethod
    def read(filename, path, library=None):
        library = library or lib()
        rank, dims = C.c_int(), (C.c_uint64 * 2)()
        library.call("rpde_h5_shape", str(filename).encode(), path.encode(), C.byref(rank), dims)
        shape = tuple(int(dims[i]) for i in range(rank.value))
        out = np.empty(shape)
        library.call("rpde_h5_read", str(filename).encode(), path.encode(), ptr(out), out.size)
        return out

    @staticmethod
    def write(filename, path, array, library=None):
        library = library or lib()
        a = as_f64(array)
        dims = (C.c_uint64 * 2)(*a.shape)
        library.call("rpde_h5_write", str(filename).encode(), path.encode(), a.ndim, dims, ptr(a))


class Navier2DMpi:
    """The reference's `Navier2DMpi` (src/navier_stokes_mpi/navier.rs:216-225, 364-373): the same constructors with the
    communicator in front -- `universe` is what the reference calls its `&Universe`, here a `rustpde_mpi_amd.dist.TorchComm`
    (callback transport over torch.distributed) or `RcclComm` (native RCCL transport).  One process per GPU; every rank
    passes the same arguments.  The object returned is a pencil-sharded `Navier2D`: `update()`, `exit()`, `callback()`,
    `write()` / `read()`, the field views and `Statistics` behave as on one GPU (collective where the reference's are)."""

    @staticmethod
    def new_confined(universe, nx, ny, ra, pr, dt, aspect, bc, device=0, library=None, init_random=0.1, seed=0):
        return Navier2D.new_confined(nx, ny, ra, pr, dt, aspect, bc, device=device, library=library, comm=universe,
                                     init_random=init_random, seed=seed)

    @staticmethod
    def new_periodic(universe, nx, ny, ra, pr, dt, aspect, bc, device=0, library=None, init_random=0.1, seed=0):
        return Navier2D.new_periodic(nx, ny, ra, pr, dt, aspect, bc, device=device, library=library, comm=universe,
                                     init_random=init_random, seed=seed)


class _StatField:
    """One member of `Statistics` (a `Field2` of the orthonormal `field` space): `.vhat` reads the coefficients."""

    def __init__(self, stats, name):
        self._stats, self._name = stats, name

    @property
    def vhat(self):
        nav = self._stats._nav
        rows = nav.nx // 2 + 1 if nav.periodic else nav.nx
        out = np.empty((rows, nav.ny * (2 if nav.periodic else 1)))
        nav._lib.call("rpde_navier2d_statistics_get", nav._h, self._name.encode(), ptr(out), out.size)
        return out.view(np.complex128) if nav.periodic else out


class Statistics:
    """`Statistics<T, S>` (src/navier_stokes/statistics.rs:11-108) kept on the device of its `Navier2D`:
    `t_avg` (running mean of temp.to_ortho()), `ux_avg` / `uy_avg` (the LAST velx / vely .to_ortho(): the
    reference assigns), `nusselt` (Nusselt field of the last snapshot), `avg_time`, `tot_time`, `num_save`.

        nav.statistics = Statistics.new(nav, save_stat=1.0, write_stat=10.0)   # then nav.callback() does the rest

    `update()` takes no arrays: it reads the fields of its engine, which is what the only call site of the
    reference passes (navier_io.rs:110-115)."""

    def __init__(self, navier, save_stat, write_stat):
        self._nav = navier
        self.save_stat, self.write_stat = float(save_stat), float(write_stat)
        # the engine keeps ONE set of statistics fields (the reference's Statistics is a value of its own,
        # statistics.rs:11-108): a second Statistics for the same solver would reset and alias the first one's data
        if getattr(navier, "_statistics_owner", None) is not None and navier._statistics_owner() is not None:
            raise RpdeError("this Navier2D already has a Statistics object; the engine keeps one set of statistics per solver")
        import weakref
        navier._statistics_owner = weakref.ref(self)
        navier._lib.call("rpde_navier2d_statistics_enable", navier._h, self.save_stat, self.write_stat)
        # Statistics::new does not touch the solver: the hook is `navier.statistics = Some(..)` (the setter above)
        navier._lib.call("rpde_navier2d_statistics_attach", navier._h, 0)
        navier._statistics = None
        self.t_avg, self.ux_avg, self.uy_avg, self.nusselt = (_StatField(self, n) for n in ("temp", "ux", "uy", "nusselt"))

    @classmethod
    def new(cls, navier, save_stat, write_stat):
        return cls(navier, save_stat, write_stat)

    def _scalars(self):
        a, t, n = C.c_double(), C.c_double(), C.c_longlong()
        self._nav._lib.call("rpde_navier2d_statistics_scalars", self._nav._h, C.byref(a), C.byref(t), C.byref(n))
        return a.value, t.value, n.value

    avg_time = property(lambda self: self._scalars()[0])
    tot_time = property(lambda self: self._scalars()[1])
    num_save = property(lambda self: self._scalars()[2])

    def update(self):
        self._nav._lib.call("rpde_navier2d_statistics_update", self._nav._h)

    def write(self, filename):
        self._nav._lib.call("rpde_navier2d_statistics_write", self._nav._h, str(filename).encode())

    def read(self, filename):
        self._nav._lib.call("rpde_navier2d_statistics_read", self._nav._h, str(filename).encode())

    def write_unwrap(self, filename):
        try:
            self.write(filename)
        except RpdeError as exc:
            print(f"Error while writing file {filename!r}. Error: {exc}")

    def read_unwrap(self, filename):
        try:
            self.read(filename)
            print(f"Reading file {filename!r} was successfull.")
        except RpdeError as exc:
            print(f"Error while reading file {filename!r}. Error: {exc}")


def microbench(what, n, nlines, reps=20, device=0, library=None):
    """Mean device time (ms) of one launch of LOAD + <what> + STORE on `nlines` lines of n points."""
    library = library or lib()
    ms = C.c_double()
    library.call("rpde_microbench", what.encode(), int(n), int(nlines), int(reps), int(device), C.byref(ms))
    return ms.value
