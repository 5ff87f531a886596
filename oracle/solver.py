"""Banded / tensor solvers of rustpde ``src/solver`` -- CPU oracle (test infrastructure).

Restates, lane-vectorised in NumPy:
  Fdma          ``src/solver/fdma.rs:33-118``      (4-diagonal solve, offsets -2,0,+2,+4)
  MatVecFdma    ``src/solver/matvec.rs:177-228``   (banded mat-vec, the B2 preconditioner)
  Sdma          ``src/solver/sdma.rs:21-46``       (diagonal solve)
  PdmaPlus2     ``src/solver/pdma_plus2.rs:45-157`` (7-diagonal solve, offsets -2 .. +4: the "hc" temperature)
  HholtzAdi     ``src/solver/hholtz_adi.rs:48-76,149-169``
  FdmaTensor    ``src/solver/fdma_tensor.rs:106-154``
  Poisson       ``src/solver/poisson.rs:54-94,195-236``
  eig / inv     ``src/solver/utils.rs:67-107``
All lane routines are written for axis 0 of a 2-D array (axis 1 = batch).
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as _la

from .bases import Base, Space2, FOURIER_R2C, CHEB_DIRICHLET_NEUMANN, _axis0
from .timing import phase, timed


# --------------------------------------------------------------------------- Fdma
def fdma_sweep(low, dia, up1, up2):
    """Forward elimination of the bands, ``fdma.rs:73-82``.  Arrays are indexed as in the
    reference (``low[i-2] = a[i, i-2]``, ``up1[i] = a[i, i+2]``, ``up2[i] = a[i, i+4]``);
    a trailing batch dimension is allowed (one matrix per batch entry)."""
    low, dia, up1, up2 = (np.array(b, dtype=np.float64, copy=True) for b in (low, dia, up1, up2))
    n = dia.shape[0]
    for i in range(2, n):
        low[i - 2] = low[i - 2] / dia[i - 2]
        dia[i] = dia[i] - low[i - 2] * up1[i - 2]
        if i < n - 2:
            up1[i] = up1[i] - low[i - 2] * up2[i - 2]
    return low, dia, up1, up2


def fdma_solve0(bands, x):
    """``Fdma::fdma`` (``fdma.rs:101-118``) along axis 0 of ``x`` (n, batch) with swept bands.
    Bands are either 1-D (same matrix for all lanes) or (len, batch)."""
    low, dia, up1, up2 = bands
    if low.ndim == 1:
        low, dia, up1, up2 = (b[:, None] for b in (low, dia, up1, up2))
    x = np.array(x, copy=True)
    n = x.shape[0]
    for i in range(2, n):
        x[i] = x[i] - x[i - 2] * low[i - 2]
    x[n - 1] = x[n - 1] / dia[n - 1]
    x[n - 2] = x[n - 2] / dia[n - 2]
    x[n - 3] = (x[n - 3] - x[n - 1] * up1[n - 3]) / dia[n - 3]
    x[n - 4] = (x[n - 4] - x[n - 2] * up1[n - 4]) / dia[n - 4]
    for i in range(n - 5, -1, -1):
        x[i] = (x[i] - x[i + 2] * up1[i] - x[i + 4] * up2[i]) / dia[i]
    return x


class Fdma:
    """Four-diagonal matrix (offsets -2, 0, +2, +4) given by row-indexed bands of length n
    (``low[r] = a[r, r-2]``, ``up1[r] = a[r, r+2]``, ``up2[r] = a[r, r+4]``; out of range = 0)."""

    def __init__(self, low, dia, up1, up2, sweep=True):
        n = len(dia)
        self.n = n
        self.raw = (np.asarray(low, float)[2:].copy(), np.asarray(dia, float).copy(),
                    np.asarray(up1, float)[: n - 2].copy(), np.asarray(up2, float)[: n - 4].copy())
        self.swept = fdma_sweep(*self.raw) if sweep else None

    def solve(self, x, axis):
        assert self.swept is not None, "Fdma: Forward sweep must be performed before solve!"
        assert x.shape[axis] == self.n, "Fdma: dimension mismatch"
        return _axis0(lambda y: fdma_solve0(self.swept, y), x, axis)


# --------------------------------------------------------------------------- MatVecFdma
class MatVecFdma:
    """Banded mat-vec with an (m x n) matrix holding offsets 0, +2, +4 (``matvec.rs:177-228``)."""

    def __init__(self, dia, up1, up2, n_in):
        self.m = len(dia)
        self.n = n_in
        m = self.m
        self.dia = np.asarray(dia, float)
        self.up1 = np.where(np.arange(m) < m - 2, up1, 0.0)
        self.up2 = np.where(np.arange(m) < m - 4, up2, 0.0)

    def _apply0(self, x):
        m = self.m
        out = x[:m] * self.dia[:, None]
        out[: m - 2] += x[2:m] * self.up1[: m - 2, None]
        out[: m - 4] += x[4:m] * self.up2[: m - 4, None]
        return out

    def apply(self, x, axis):
        assert x.shape[axis] == self.n
        return _axis0(self._apply0, x, axis)


# --------------------------------------------------------------------------- Sdma
class Sdma:
    def __init__(self, dia):
        self.dia = np.asarray(dia, float)
        self.n = len(dia)

    def solve(self, x, axis):
        shape = [1, 1]
        shape[axis] = self.n
        return x / self.dia.reshape(shape)


# --------------------------------------------------------------------------- PdmaPlus2
class PdmaPlus2:
    """Banded system with offsets -2 .. +4 (``pdma_plus2.rs``): LU without pivoting, the factorisation is
    precomputed (``from_matrix``, ``:45-117``), a solve is a two-term forward and a four-term backward recurrence
    (``solve_lane``, ``:119-157``).  ``bands``: offset -> array of length n, entry r = a[r, r + offset] (row
    indexed, zero where the column is out of range) -- the reference extracts the same numbers with ``diag(a, k)``
    (column-indexed for the lower diagonals: ``l2[i] = a[i+2, i]``, ``l1[i] = a[i+1, i]``).  Written with
    zero-padded arrays, so the reference's special cases for the first two and last four rows are the general
    row with vanishing out-of-range terms, in the reference's order of operations."""

    def __init__(self, bands):
        n = len(bands[0])
        self.n = n
        z = lambda: np.zeros(n + 4)
        l2, l1, d0, u1, u2, u3, u4 = z(), z(), z(), z(), z(), z(), z()
        l2[: n - 2] = np.asarray(bands[-2], float)[2:]        # l2[i] = a[i+2, i]
        l1[: n - 1] = np.asarray(bands[-1], float)[1:]        # l1[i] = a[i+1, i]
        d0[:n] = bands[0]
        for arr, o in ((u1, 1), (u2, 2), (u3, 3), (u4, 4)):
            arr[: n - o] = np.asarray(bands[o], float)[: n - o]
        al, be, ga, de, ka, mu = z(), z(), z(), z(), z(), z()
        for i in range(n):
            a2 = al[i - 2] if i >= 2 else 0.0
            b2 = be[i - 2] if i >= 2 else 0.0
            g2 = ga[i - 2] if i >= 2 else 0.0
            e2 = de[i - 2] if i >= 2 else 0.0
            w2 = l2[i - 2] if i >= 2 else 0.0
            a1 = al[i - 1] if i >= 1 else 0.0
            b1 = be[i - 1] if i >= 1 else 0.0
            g1 = ga[i - 1] if i >= 1 else 0.0
            e1 = de[i - 1] if i >= 1 else 0.0
            ka[i] = (l1[i - 1] if i >= 1 else 0.0) - a2 * w2
            mu[i] = d0[i] - b2 * w2 - a1 * ka[i]
            al[i] = (u1[i] - g2 * w2 - b1 * ka[i]) / mu[i]
            be[i] = (u2[i] - e2 * w2 - g1 * ka[i]) / mu[i]
            ga[i] = (u3[i] - e1 * ka[i]) / mu[i]
            de[i] = u4[i] / mu[i]
        self.al, self.be, self.ga, self.de, self.l2, self.ka, self.mu = al, be, ga, de, l2, ka, mu

    def _solve0(self, rhs):
        n = self.n
        l2, ka, mu = self.l2, self.ka, self.mu
        ze = np.zeros((n + 4,) + rhs.shape[1:], dtype=rhs.dtype)
        for i in range(n):
            acc = rhs[i]
            if i >= 2:
                acc = acc - ze[i - 2] * l2[i - 2]
            if i >= 1:
                acc = acc - ze[i - 1] * ka[i]
            ze[i] = acc / mu[i]
        x = ze                      # entries n .. n+3 stay zero: the four-term recurrence needs no end cases
        al, be, ga, de = self.al, self.be, self.ga, self.de
        for i in range(n - 2, -1, -1):
            x[i] = ze[i] - x[i + 1] * al[i] - x[i + 2] * be[i] - x[i + 3] * ga[i] - x[i + 4] * de[i]
        return x[:n].copy()

    def solve(self, x, axis):
        assert x.shape[axis] == self.n, "PdmaPlus2: dimension mismatch"
        return _axis0(self._solve0, x, axis)


# --------------------------------------------------------------------------- ingredients
def ingredients_for_hholtz(base: Base):
    """``field.rs:195-216``: band forms of (mat_a, mat_b, precond)."""
    if base.kind == FOURIER_R2C:
        k = base.wavenumbers()
        return ("diag", np.ones(base.m)), ("diag", -(k ** 2)), None
    if not base.is_composite:
        raise NotImplementedError("solver ingredients for the orthonormal Chebyshev base "
                                  "are not on the Navier2D path")
    if base.kind == CHEB_DIRICHLET_NEUMANN:
        mat_a, mat_b = base.hholtz_bands7()
        return ("band7", mat_a), ("band7", mat_b), base.pinv_bands()
    mat_a, mat_b = base.hholtz_bands()
    return ("band", mat_a), ("band", mat_b), base.pinv_bands()


# --------------------------------------------------------------------------- HholtzAdi
class HholtzAdi:
    """(I - c D2) vhat = A f, ADI-factored per axis   (``hholtz_adi.rs:48-76``)."""

    def __init__(self, space: Space2, c):
        self.solver = []
        self.matvec = []
        for axis, ci in enumerate(c):
            base = space.bases[axis]
            (ka, a), (_, b), precond = ingredients_for_hholtz(base)
            if ka == "diag":
                self.solver.append(Sdma(a - b * ci))
                self.matvec.append(None)
            elif ka == "band7":       # BaseKind::ChebDirichletNeumann => PdmaPlus2 (hholtz_adi.rs:62-64)
                self.solver.append(PdmaPlus2({o: a[o] - b.get(o, 0.0) * ci for o in a}))
                self.matvec.append(MatVecFdma(*precond, n_in=base.n))
            else:
                a_low, a_dia, a_up1, a_up2 = a
                b_dia, b_up1 = b
                self.solver.append(Fdma(a_low, a_dia - b_dia * ci, a_up1 - b_up1 * ci, a_up2))
                self.matvec.append(MatVecFdma(*precond, n_in=base.n))

    @timed("helmholtz")
    def solve(self, inp):
        rhs = inp
        for axis in (0, 1):
            if self.matvec[axis] is not None:
                rhs = self.matvec[axis].apply(rhs, axis)
        out = self.solver[0].solve(rhs, 0)
        return self.solver[1].solve(out, 1)


class HholtzAdi1:
    """1-D variant (``hholtz_adi.rs:78-118``), used by the reference's known-answer test."""

    def __init__(self, base: Base, c):
        self._h = HholtzAdi.__new__(HholtzAdi)
        (ka, a), (_, b), precond = ingredients_for_hholtz(base)
        if ka == "band7":
            self.solver = PdmaPlus2({o: a[o] - b.get(o, 0.0) * c for o in a})
        else:
            a_low, a_dia, a_up1, a_up2 = a
            b_dia, b_up1 = b
            self.solver = Fdma(a_low, a_dia - b_dia * c, a_up1 - b_up1 * c, a_up2)
        self.matvec = MatVecFdma(*precond, n_in=base.n)

    def solve(self, b):
        return self.solver.solve(self.matvec.apply(np.asarray(b, float)[:, None], 0), 0)[:, 0]


# --------------------------------------------------------------------------- eig helpers
def eig_sorted(xmat):
    """``utils.rs:67-99``: real parts of LAPACK dgeev, sorted largest -> smallest, plus inverse."""
    lam, q = _la.eig(xmat)
    lam = lam.real
    q = q.real
    perm = np.argsort(lam, kind="stable")[::-1]
    lam = lam[perm]
    q = np.ascontiguousarray(q[:, perm])
    return lam, q, _la.inv(q)


def band_to_dense(low, dia, up1, up2):
    n = len(dia)
    a = np.zeros((n, n))
    for r in range(n):
        a[r, r] = dia[r]
        if r >= 2:
            a[r, r - 2] = low[r]
        if r + 2 < n:
            a[r, r + 2] = up1[r]
        if r + 4 < n:
            a[r, r + 4] = up2[r]
    return a


def eigen_decomposition_x(a_bands, c_bands, mode="full"):
    """Diagonalise inv(C) A along the inner axis (``fdma_tensor.rs:123-127``).

    mode="full"   : exactly the reference (one dense dgeev of the (m x m) matrix).
    mode="parity" : even and odd coefficients decouple (all bands have even offsets);
                    diagonalise the two blocks separately.  Same discrete operator, same
                    solution up to round-off; 4x cheaper and immune to LAPACK mixing
                    near-degenerate even/odd pairs.
    Returns lam (m), fwd = Q^-1 C^-1 (m x m), bwd = Q (m x m)."""
    a = band_to_dense(*a_bands)
    c = band_to_dense(*c_bands)
    m = a.shape[0]
    if mode == "full":
        cinv = _la.inv(c)
        lam, q, qinv = eig_sorted(cinv @ a)
        return lam, qinv @ cinv, q
    lam = np.zeros(m)
    fwd = np.zeros((m, m))
    bwd = np.zeros((m, m))
    pos = 0
    blocks = []
    for par in (0, 1):
        idx = np.arange(par, m, 2)
        cinv = _la.inv(c[np.ix_(idx, idx)])
        l, q, qinv = eig_sorted(cinv @ a[np.ix_(idx, idx)])
        blocks.append((idx, l, q, qinv @ cinv))
    # interleave the two spectra into one descending list (as a full sort would)
    all_l = np.concatenate([b[1] for b in blocks])
    owner = np.concatenate([np.full(len(b[1]), p) for p, b in enumerate(blocks)])
    local = np.concatenate([np.arange(len(b[1])) for b in blocks])
    perm = np.argsort(all_l, kind="stable")[::-1]
    for pos, j in enumerate(perm):
        idx, l, q, f = blocks[owner[j]]
        lam[pos] = l[local[j]]
        bwd[idx, pos] = q[:, local[j]]
        fwd[pos, idx] = f[local[j], :]
    return lam, fwd, bwd


# --------------------------------------------------------------------------- Poisson
class Poisson:
    """c D2 vhat = A f via eigen-decomposition in x and banded row solves in y
    (``poisson.rs:54-94,195-236``; tensor data ``fdma_tensor.rs:74-154``)."""

    # (the tensor Helmholtz solver of hholtz.rs is the same construction with other constants: class Hholtz below)
    LAPLACIAN_SIGN = 1.0      # laplacian = +mat_b * c          (poisson.rs:68)
    ALPHA = 0.0               # FdmaTensor::from_matrix(.., 0.) (poisson.rs:82)
    SINGULARITY_FIX = True    # poisson.rs:84-87

    def __init__(self, space: Space2, c, eig_mode="full", eig_override=None):
        c = [self.LAPLACIAN_SIGN * ci for ci in c]
        b0, b1 = space.bases
        self.matvec = []
        ing = []
        for axis, ci in enumerate(c):
            base = space.bases[axis]
            (ka, a), (_, b), precond = ingredients_for_hholtz(base)
            self.matvec.append(None if precond is None else MatVecFdma(*precond, n_in=base.n))
            if ka == "diag":
                ing.append(("diag", a, b * ci))
            else:
                b_dia, b_up1 = b
                z = np.zeros_like(b_dia)
                ing.append(("band", a, (z, b_dia * ci, b_up1 * ci, z)))  # (mass, laplacian)
        # inner axis (0)
        kind0, mass0, lap0 = ing[0]
        if kind0 == "diag":
            self.lam = lap0.copy()
            self.fwd = self.bwd = None
        elif eig_override is not None:
            self.lam, self.fwd, self.bwd = (np.array(x, copy=True) for x in eig_override)
        else:
            self.lam, self.fwd, self.bwd = eigen_decomposition_x(lap0, mass0, eig_mode)
        # outermost axis (1): raw (unswept) bands of A_y = laplacian, C_y = mass
        kind1, mass1, lap1 = ing[1]
        assert kind1 == "band"
        self.fdma_a = Fdma(*lap1, sweep=False)
        self.fdma_c = Fdma(*mass1, sweep=False)
        self.alpha = self.ALPHA
        # singularity fix, poisson.rs:84-87 (shifts the WHOLE eigenvalue vector)
        if self.SINGULARITY_FIX and abs(self.lam[0]) < 1e-10:
            self.lam = self.lam - 1e-10

    def row_bands(self):
        """Swept bands of A_y + lam_i C_y for every x-row i: arrays (len, nrows)."""
        l = (self.lam + self.alpha)[None, :]
        raw = [a[:, None] + c[:, None] * l for a, c in zip(self.fdma_a.raw, self.fdma_c.raw)]
        return fdma_sweep(*raw)

    def solve(self, inp):
        rhs = inp
        with phase("poisson_rows"):      # preconditioner + per-row factorisation and solve (poisson.rs:205-229)
            for axis in (0, 1):
                if self.matvec[axis] is not None:
                    rhs = self.matvec[axis].apply(rhs, axis)
        with phase("poisson_gemm"):      # poisson.rs:214-216
            out = self.fwd @ rhs if self.fwd is not None else rhs
        with phase("poisson_rows"):
            # rows i of `out` are lanes along y; batch them: (ny_m, nx_m)
            out = np.ascontiguousarray(fdma_solve0(self.row_bands(), np.ascontiguousarray(out.T)).T)
        with phase("poisson_gemm"):      # poisson.rs:232-235
            if self.bwd is not None:
                out = self.bwd @ out
        return out


class Hholtz(Poisson):
    """(I - c D2) vhat = A f with the tensor solver (``src/solver/hholtz.rs:72-106``): the construction of ``Poisson`` with
    laplacian = -mat_b * c (``hholtz.rs:86``), alpha = 1 (``hholtz.rs:100``: the identity becomes alpha * (Cx x Cy)) and no
    singularity fix; ``solve`` = preconditioner along both axes, then ``FdmaTensor::solve`` (``hholtz.rs:176-186``).
    Used by ``Navier2DAdjoint`` as the norm of the residual (``steady_adjoint.rs:296-318``)."""
    LAPLACIAN_SIGN = -1.0
    ALPHA = 1.0
    SINGULARITY_FIX = False


class Poisson1:
    """1-D variant (``poisson.rs:96-140``; ``fdma_tensor.rs:148-152``: A swept once, alpha = 0)."""

    def __init__(self, base: Base, c):
        (ka, a), (_, b), precond = ingredients_for_hholtz(base)
        b_dia, b_up1 = b
        z = np.zeros_like(b_dia)
        self.solver = Fdma(z, b_dia * c, b_up1 * c, z)
        self.matvec = MatVecFdma(*precond, n_in=base.n)

    def solve(self, b):
        return self.solver.solve(self.matvec.apply(np.asarray(b, float)[:, None], 0), 0)[:, 0]
