"""Function spaces of funspace 0.3.0 as used by rustpde -- CPU oracle (test infrastructure).

funspace is an external crate (``Cargo.toml:17``, ``Cargo.lock:451-463`` of the
reference); its source is not in ``/root/reference``.  This file restates its
published behaviour at the call sites the reference
uses: ``src/bases.rs:11-19`` (re-exports), ``src/field.rs:85-88,103-129,195-216``.

Everything here works on 2-D row-major arrays and acts along one axis
("lanes"), exactly like funspace's ``*_par`` lane iterators.
"""
from __future__ import annotations

import numpy as np
import scipy.fft as _fft

WORKERS = -1  # scipy.fft worker threads (rayon lanes in the reference)
from .timing import timed  # noqa: E402  (per-phase CPU baseline of bench.py)

CHEBYSHEV = "chebyshev"
CHEB_DIRICHLET = "cheb_dirichlet"
CHEB_NEUMANN = "cheb_neumann"
CHEB_DIRICHLET_NEUMANN = "cheb_dirichlet_neumann"
FOURIER_R2C = "fourier_r2c"


def _axis0(fn, x, axis, *a, **k):
    """Run lane function ``fn`` (written for axis 0) along ``axis``.  (Round 5 tried blocks of lanes on a thread pool:
    bit-identical, 10 % faster in `differentiate` on 64 cores, and five times SLOWER in the Helmholtz sweeps, whose lane
    functions are Python loops over the line -- GIL-bound; not kept.)"""
    if axis == 0:
        return fn(x, *a, **k)
    return np.ascontiguousarray(fn(np.ascontiguousarray(x.T), *a, **k).T)


class Base:
    """One 1-D basis (funspace ``BaseR2r`` / ``BaseR2c``).

    kind: chebyshev | cheb_dirichlet | cheb_neumann | cheb_dirichlet_neumann | fourier_r2c
    n   : number of grid points (physical size)
    m   : number of spectral coefficients (n, n-2, or n//2+1 complex)
    """

    def __init__(self, kind: str, n: int):
        self.kind = kind
        self.n = n
        if kind == CHEBYSHEV:
            self.m = n
        elif kind in (CHEB_DIRICHLET, CHEB_NEUMANN, CHEB_DIRICHLET_NEUMANN):
            self.m = n - 2
        elif kind == FOURIER_R2C:
            self.m = n // 2 + 1
        else:
            raise ValueError(f"unknown base kind {kind!r}")
        if self.is_composite:
            k = np.arange(self.m, dtype=np.float64)
            # stencil S (n x m): S[k,k] = dia[k], S[k+2,k] = low[k]   (App. A.3)
            self.dia = np.ones(self.m)
            self.low1 = None     # S[k+1,k]: only the three-term stencil of cheb_dirichlet_neumann has it
            if kind == CHEB_DIRICHLET:
                self.low = -np.ones(self.m)
            elif kind == CHEB_NEUMANN:
                self.low = -((k / (k + 2.0)) ** 2)
            else:
                # phi_k = T_k + a_k T_{k+1} + b_k T_{k+2} with phi_k(-1) = 0 and phi_k'(+1) = 0 (Shen's mixed basis;
                # T_k(-1) = (-1)^k, T_k'(1) = k^2):  b_k = a_k - 1,  k^2 + a_k (k+1)^2 + b_k (k+2)^2 = 0.
                # The wall assignment follows the lift of the reference: bc_hc (boundary_conditions.rs:96-134)
                # carries the bottom temperature at y[0] = -1 and has T = T' = 0 at y[n-1] = +1, so the
                # homogeneous part is Dirichlet at -1 and Neumann at +1 (navier.rs:245-248).
                den = (k + 1.0) ** 2 + (k + 2.0) ** 2
                self.low1 = 4.0 * (k + 1.0) / den
                self.low = -(k ** 2 + (k + 1.0) ** 2) / den
            # normal equations (S^T S) a = S^T c : offsets -2, 0, +2 (and -1, +1 for the three-term stencil)
            self.ls_main = self.dia ** 2 + self.low ** 2
            self.ls_off = self.dia[2:] * self.low[:-2]
            if self.low1 is not None:
                self.ls_main = self.ls_main + self.low1 ** 2
                self.ls_off1 = self.low1[:-1] * self.dia[1:] + self.low[:-1] * self.low1[1:]

    # ------------------------------------------------------------------ kinds
    @property
    def is_composite(self):
        return self.kind in (CHEB_DIRICHLET, CHEB_NEUMANN, CHEB_DIRICHLET_NEUMANN)

    @property
    def is_cheb(self):
        return self.kind != FOURIER_R2C

    @property
    def spectral_dtype(self):
        return np.complex128 if self.kind == FOURIER_R2C else np.float64

    # ------------------------------------------------------------------ grid
    def coords(self):
        """Grid points (App. A.1): Gauss-Lobatto ascending / 2 pi j / n."""
        j = np.arange(self.n, dtype=np.float64)
        if self.is_cheb:
            return -np.cos(np.pi * j / (self.n - 1))
        return 2.0 * np.pi * j / self.n

    # ------------------------------------------------- orthonormal transforms
    def _cheb_fwd_factor(self):
        n = self.n
        f = np.where(np.arange(n) % 2 == 0, 1.0, -1.0) / (n - 1)
        f[0] *= 0.5
        f[-1] *= 0.5
        return f

    def _cheb_bwd_factor(self):
        n = self.n
        f = np.where(np.arange(n) % 2 == 0, 1.0, -1.0) * 0.5
        f[0] *= 2.0
        f[-1] *= 2.0
        return f

    @timed("transforms")
    def forward_ortho(self, v, axis):
        """physical (n) -> ORTHO coefficients (n real | n/2+1 complex)   (App. A.2, A.5)."""
        if self.is_cheb:
            y = _fft.dct(v, type=1, axis=axis, workers=WORKERS)
            shape = [1, 1]
            shape[axis] = self.n
            return y * self._cheb_fwd_factor().reshape(shape)
        return _fft.rfft(v, axis=axis, workers=WORKERS)  # unnormalised forward

    @timed("transforms")
    def backward_ortho(self, c, axis):
        """ORTHO coefficients -> physical (n)."""
        if self.is_cheb:
            shape = [1, 1]
            shape[axis] = self.n
            return _fft.dct(c * self._cheb_bwd_factor().reshape(shape), type=1, axis=axis,
                            workers=WORKERS)
        return _fft.irfft(c, n=self.n, axis=axis, workers=WORKERS)  # carries 1/n

    # ------------------------------------------------- composite <-> ortho
    @timed("stencil_gradient")
    def to_ortho(self, a, axis):
        """composite (m) -> ortho (n): c = S a   (``field.rs:113-115``)."""
        if not self.is_composite:
            return a.copy()
        return _axis0(self._to_ortho0, a, axis)

    def _to_ortho0(self, a):
        c = np.zeros((self.n,) + a.shape[1:], dtype=a.dtype)
        c[: self.m] = self.dia[:, None] * a
        c[2:] += self.low[:, None] * a
        if self.low1 is not None:
            c[1: self.m + 1] += self.low1[:, None] * a
        return c

    @timed("stencil_gradient")
    def from_ortho(self, c, axis):
        """ortho (n) -> composite (m): least squares (S^T S) a = S^T c, stride-2 TDMA."""
        if not self.is_composite:
            return c.copy()
        return _axis0(self._from_ortho0, c, axis)

    def _from_ortho0(self, c):
        m = self.m
        d = self.dia[:, None] * c[:m] + self.low[:, None] * c[2:]
        if self.low1 is not None:
            # three-term stencil: S^T S is pentadiagonal and symmetric positive definite; any exact solver gives
            # the least-squares projection (the algorithm funspace uses for it is not visible from the reference)
            import scipy.linalg as _la
            d = d + self.low1[:, None] * c[1: m + 1]
            ab = np.zeros((3, m))
            ab[0] = self.ls_main
            ab[1, : m - 1] = self.ls_off1
            ab[2, : m - 2] = self.ls_off
            return _la.solveh_banded(ab, d, lower=True)
        off, main = self.ls_off, self.ls_main
        # forward sweep (data independent part)
        w = np.zeros(m)
        den = np.zeros(m)
        den[:2] = main[:2]
        for i in range(m):
            if i >= 2:
                den[i] = main[i] - off[i - 2] * w[i - 2]
            if i < m - 2:
                w[i] = off[i] / den[i]
        g = np.empty_like(d)
        g[0] = d[0] / den[0]
        g[1] = d[1] / den[1]
        for i in range(2, m):
            g[i] = (d[i] - off[i - 2] * g[i - 2]) / den[i]
        x = g
        for i in range(m - 3, -1, -1):
            x[i] = g[i] - w[i] * x[i + 2]
        return x

    # ------------------------------------------------------------ transforms
    def forward(self, v, axis):
        """physical -> this base's coefficients (composite if the base is composite)."""
        c = self.forward_ortho(v, axis)
        return self.from_ortho(c, axis) if self.is_composite else c

    def backward(self, a, axis):
        return self.backward_ortho(self.to_ortho(a, axis), axis)

    # ------------------------------------------------------- differentiation
    @timed("stencil_gradient")
    def differentiate(self, a, order, axis):
        """coefficients -> ORTHO coefficients of the ``order``-th derivative (App. A.4).

        Always converts to the orthonormal parent first (also for order 0)."""
        c = self.to_ortho(a, axis)
        if order == 0:
            return c
        if self.is_cheb:
            return _axis0(self._cheb_diff0, c, axis, order)
        k = np.arange(self.m, dtype=np.float64)
        fac = (1j * k) ** order
        shape = [1, 1]
        shape[axis] = self.m
        return c * fac.reshape(shape)

    def _cheb_diff0(self, c, order):
        n = self.n
        for _ in range(order):
            d = np.zeros_like(c)
            # d_k = d_{k+2} + 2 (k+1) c_{k+1},  k = n-2 .. 1 ;  d_0 = c_1 + d_2 / 2
            t = (2.0 * np.arange(1, n, dtype=np.float64))[:, None] * c[1:]  # t[k] = 2(k+1)c[k+1]
            # two independent stride-2 chains, accumulated from the top down
            for start in (n - 2, n - 3):
                idx = np.arange(start, 0, -2)  # descending k >= 1 of one parity
                if idx.size:
                    d[idx] = np.cumsum(t[idx], axis=0)
            d[0] = c[1] + (d[2] / 2.0 if n > 2 else 0.0)
            c = d
        return c

    # ------------------------------------------------ matrices for the solvers
    def stencil_bands(self):
        """(dia, low) of S for composite bases; identity for ortho."""
        if self.is_composite:
            return self.dia, self.low
        raise ValueError("no stencil for an orthonormal base")

    def pinv_bands(self):
        """Rows 2.. of the D2 pseudo-inverse B2 (App. A.6): (m x n), offsets 0, +2, +4.

        ``peye . laplace_inv`` of ``src/field.rs:203-210``; the returned arrays have
        length n-2 and hold pinv[r, r], pinv[r, r+2], pinv[r, r+4]."""
        n = self.n
        r = np.arange(n - 2)
        i = (r + 2).astype(np.float64)
        d0 = 1.0 / (4.0 * i * (i - 1.0))
        d0[0] = 0.25
        d2 = np.where(r + 2 <= n - 3, -1.0 / (2.0 * (i * i - 1.0)), 0.0)
        d4 = np.where(r + 2 <= n - 5, 1.0 / (4.0 * i * (i + 1.0)), 0.0)
        return d0, d2, d4

    def hholtz_bands7(self):
        """(mat_a, mat_b) = (pinv.S, peye.S) of ``field.rs:208-212`` for any composite stencil as dictionaries
        offset -> array of length m (entry r = mat[r, r + offset], zero where the column is out of range):
        mat_a has offsets -2 .. +4, mat_b offsets 0 .. +2.  With a two-term stencil the odd offsets vanish and the
        rest equals ``hholtz_bands``; the three-term stencil of cheb_dirichlet_neumann fills all seven -- the
        matrix ``PdmaPlus2`` solves (``hholtz_adi.rs:62-64``, ``pdma_plus2.rs:45-53``)."""
        assert self.is_composite
        m = self.m
        p = dict(zip((0, 2, 4), self.pinv_bands()))
        low1 = self.low1 if self.low1 is not None else np.zeros(m)
        sten = {0: self.dia, 1: low1, 2: self.low}          # sten[t][j] = S[j + t, j]
        r = np.arange(m)

        def s(t, j):          # S[j + t, j] for an index array j, zero outside 0 <= j < m
            ok = (j >= 0) & (j < m)
            return np.where(ok, sten[t][np.clip(j, 0, m - 1)], 0.0)

        mat_a = {o: sum(p[q] * s(q - o, r + o) for q in (0, 2, 4) if 0 <= q - o <= 2) for o in range(-2, 5)}
        mat_b = {o: s(2 - o, r + o) for o in (0, 1, 2)}     # peye[r, r+2] = 1
        return mat_a, mat_b

    def hholtz_bands(self):
        """Band forms of (mat_a, mat_b) = (pinv.S, peye.S) of ``field.rs:208-212``.

        mat_a: offsets (-2, 0, +2, +4) -> arrays (low, dia, up1, up2) of length m
               (entry r of ``low`` is mat_a[r, r-2]; out-of-range entries are 0)
        mat_b: offsets (0, +2)         -> arrays (dia, up1) of length m
        """
        assert self.is_composite
        m = self.m
        p0, p2, p4 = self.pinv_bands()
        dia, low = self.dia, self.low
        pad = lambda x, k: np.concatenate([x[k:], np.zeros(k)])  # x[r+k]
        a_low = np.zeros(m)
        a_low[2:] = p0[2:] * low[:-2]
        a_dia = p0 * dia + p2 * low
        a_up1 = p2 * pad(dia, 2) + p4 * pad(low, 2)
        a_up2 = p4 * pad(dia, 4)
        b_dia = low.copy()
        b_up1 = pad(dia, 2)
        return (a_low, a_dia, a_up1, a_up2), (b_dia, b_up1)

    # dense forms (small n only; used to pin the band forms in the tests)
    def mass_dense(self):
        if not self.is_composite:
            return np.eye(self.m)
        s = np.zeros((self.n, self.m))
        for k in range(self.m):
            s[k, k] = self.dia[k]
            s[k + 2, k] = self.low[k]
            if self.low1 is not None:
                s[k + 1, k] = self.low1[k]
        return s

    def laplace_inv_dense(self):
        n = self.n
        b2 = np.zeros((n, n))
        b2[2, 0] = 0.25
        for i in range(3, n):
            b2[i, i - 2] = 1.0 / (4.0 * i * (i - 1))
        for i in range(2, n - 2):
            b2[i, i] = -1.0 / (2.0 * (i * i - 1))
        for i in range(2, n - 4):
            b2[i, i + 2] = 1.0 / (4.0 * i * (i + 1))
        return b2

    def laplace_inv_eye_dense(self):
        n = self.n
        e = np.zeros((n - 2, n))
        for i in range(n - 2):
            e[i, i + 2] = 1.0
        return e

    def wavenumbers(self):
        assert self.kind == FOURIER_R2C
        return np.arange(self.m, dtype=np.float64)


def chebyshev(n):
    return Base(CHEBYSHEV, n)


def cheb_dirichlet(n):
    return Base(CHEB_DIRICHLET, n)


def cheb_neumann(n):
    return Base(CHEB_NEUMANN, n)


def cheb_dirichlet_neumann(n):
    return Base(CHEB_DIRICHLET_NEUMANN, n)


def fourier_r2c(n):
    return Base(FOURIER_R2C, n)


class Space2:
    """funspace ``Space2<B0, B1>``: tensor product of two 1-D bases."""

    def __init__(self, base0: Base, base1: Base):
        self.bases = (base0, base1)

    def base_kind(self, axis):
        return self.bases[axis].kind

    @property
    def shape_physical(self):
        return (self.bases[0].n, self.bases[1].n)

    @property
    def shape_spectral(self):
        return (self.bases[0].m, self.bases[1].m)

    @property
    def shape_ortho(self):
        b0, b1 = self.bases
        return (b0.m if b0.kind == FOURIER_R2C else b0.n, b1.n)

    @property
    def spectral_dtype(self):
        return self.bases[0].spectral_dtype

    def ndarray_physical(self):
        return np.zeros(self.shape_physical)

    def ndarray_spectral(self):
        return np.zeros(self.shape_spectral, dtype=self.spectral_dtype)

    def coords(self):
        return [b.coords() for b in self.bases]

    def forward(self, v):
        """axis 1 first, then axis 0 (r2c needs real input)   (App. A.5)."""
        b0, b1 = self.bases
        return b0.forward(b1.forward(v, 1), 0)

    def backward(self, vhat):
        b0, b1 = self.bases
        return b1.backward(b0.backward(vhat, 0), 1)

    def to_ortho(self, vhat):
        b0, b1 = self.bases
        return b1.to_ortho(b0.to_ortho(vhat, 0), 1)

    def from_ortho(self, c):
        b0, b1 = self.bases
        return b1.from_ortho(b0.from_ortho(c, 0), 1)

    def gradient(self, vhat, deriv, scale=None):
        """``gradient_par`` (``field.rs:127-129``): returns ORTHO-shaped coefficients."""
        b0, b1 = self.bases
        out = b1.differentiate(b0.differentiate(vhat, deriv[0], 0), deriv[1], 1)
        if scale is not None:
            out = out / (scale[0] ** deriv[0] * scale[1] ** deriv[1])
        return out
