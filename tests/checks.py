"""Parity checks shared by the CPU (emulation build) and GPU (HIP build) test modules.

Every check drives the library through the C ABI (ctypes, rustpde_mpi_amd) and compares with the
oracle on the same seeded inputs.  Tolerances are relative L2 in f64 and are written next to
each check; the step-level bar of BASELINE.json is 1e-10.
"""
import json
import os
import subprocess
import sys

import numpy as np

import rustpde_mpi_amd as R
from oracle import bases as B, navier as N, solver as S

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = {0: "chebyshev", 1: "cheb_dirichlet", 2: "cheb_neumann", 3: "fourier_r2c", 4: "cheb_dirichlet_neumann"}
KINDS = {v: k for k, v in NAMES.items()}


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def spaces(lib, k0, n0, k1, n1):
    sp = R.Space2((KINDS[k0], n0), (KINDS[k1], n1), library=lib)
    osp = B.Space2(B.Base(k0, n0), B.Base(k1, n1))
    return sp, osp


def known_answers():
    with open(os.path.join(GOLDEN, "reference_known_answers.json")) as f:
        return json.load(f)


def check_space_ops(lib, k0, n0, k1, n1, seed=0, tol=2e-12):
    """forward / backward / to_ortho / from_ortho / gradient vs the oracle (src/field.rs:103-129)."""
    sp, osp = spaces(lib, k0, n0, k1, n1)
    rng = np.random.default_rng(seed)
    v = rng.standard_normal(osp.shape_physical)
    vh = osp.forward(v)
    assert sp.shape("physical")[:2] == osp.shape_physical
    assert sp.shape("spectral")[:2] == osp.shape_spectral
    assert sp.shape("ortho")[:2] == osp.shape_ortho
    assert rel(sp.forward(v), vh) < tol
    assert rel(sp.backward(vh), osp.backward(vh)) < tol
    c = osp.to_ortho(vh)
    assert rel(sp.to_ortho(vh), c) < tol
    assert rel(sp.from_ortho(c), osp.from_ortho(c)) < tol
    for d, sc in (([1, 0], [2.0, 1.0]), ([0, 1], [2.0, 1.0]), ([2, 0], None), ([0, 2], [1.0, 0.5]), ([1, 1], None)):
        assert rel(sp.gradient(vh, d, sc), osp.gradient(vh, d, sc)) < tol, d
    # forward(backward(x)) round trip, a size-independent property
    assert rel(sp.forward(sp.backward(vh)), vh) < tol


def check_solvers(lib, k0, n0, k1, n1, c, seed=1, tol=1e-11, eig_mode="parity", poisson_tol=None):
    """HholtzAdi / Poisson vs the oracle.  eig_mode "parity": the parity-block eigenbasis on both
    sides; "full": the oracle diagonalises the whole x operator with one dgeev like the reference
    (src/solver/utils.rs:67-99) while the engine keeps its parity blocks -- same discrete solution,
    two different LAPACK eigenbases."""
    sp, osp = spaces(lib, k0, n0, k1, n1)
    rng = np.random.default_rng(seed)
    rhs = rng.standard_normal(osp.shape_ortho)
    if k0 == "fourier_r2c":
        rhs = rhs + 1j * rng.standard_normal(osp.shape_ortho)
    e_h = rel(R.HholtzAdi(sp, c).solve(rhs), S.HholtzAdi(osp, c).solve(rhs))
    assert e_h < tol, ("HholtzAdi", e_h)
    pois = R.Poisson(sp, c)
    if eig_mode == "shared" and k0 != "fourier_r2c":
        opois = S.Poisson(osp, c, eig_override=pois.eigenbasis())
    else:
        opois = S.Poisson(osp, c, eig_mode="parity" if eig_mode == "shared" else eig_mode)
    e_p = rel(pois.solve(rhs), opois.solve(rhs))
    assert e_p < (poisson_tol or tol), ("Poisson", e_p)
    return e_h, e_p


def check_eigenbasis_is_valid(lib, n0, n1, c=(1.0, 1.0), tol=1e-8):
    """The exported (lam, fwd = Q^-1 C^-1, bwd = Q) is an eigen-decomposition of inv(C) A of the
    reference's Poisson::new (fdma_tensor.rs:123-127): C bwd diag(lam) fwd C = A, fwd C bwd = I,
    real spectrum <= 0 with one zero eigenvalue (Neumann), parity blocks decoupled."""
    sp, osp = spaces(lib, "cheb_neumann", n0, "cheb_neumann", n1)
    pois = R.Poisson(sp, list(c))
    lam, fwd, bwd = pois.eigenbasis()
    base = osp.bases[0]
    (_, a), (_, b), _ = S.ingredients_for_hholtz(base)
    b_dia, b_up1 = b
    z = np.zeros_like(b_dia)
    A = S.band_to_dense(z, b_dia * c[0], b_up1 * c[0], z)
    Cm = S.band_to_dense(*a)
    m = n0 - 2
    assert lam.shape == (m,) and abs(lam.max()) < 1e-9 and (lam <= 1e-9).all()
    # inv(C) A Q = Q diag(lam) and fwd C bwd = I.  (cond(C) grows like n^4 -- 4e12 at n = 1025 -- so
    # identities that multiply by C twice are not usable as a test; the reference forms inv(C)
    # explicitly, fdma_tensor.rs:123, and inherits that conditioning.)
    X = np.linalg.solve(Cm, A)
    assert rel(X @ bwd, bwd * lam[None, :]) < tol
    assert np.abs(fwd @ Cm @ bwd - np.eye(m)).max() < 100 * tol
    # parity structure: eigenvector k of the even block has no odd coefficients and vice versa
    me = (m + 1) // 2
    assert np.abs(bwd[1::2, :me]).max() == 0.0 and np.abs(bwd[0::2, me:]).max() == 0.0
    # and the same solve through the oracle on this decomposition agrees tightly
    rhs = np.random.default_rng(4).standard_normal(osp.shape_ortho)
    want = S.Poisson(osp, list(c), eig_override=(lam, fwd, bwd)).solve(rhs)
    e = rel(pois.solve(rhs), want)
    assert e < 1e-11, e


def check_reference_known_answers(lib):
    """The reference's own pypde vectors, through the device operators (tolerance 1e-3 absolute as
    in src/solver/hholtz_adi.rs:184 and src/solver/poisson.rs:253)."""
    g = known_answers()
    t = g["hholtz_adi_2d"]
    sp = R.Space2((1, 7), (1, 7), library=lib)
    x = R.HholtzAdi(sp, t["c"]).solve(np.tile(np.array(t["b_row"], float), (7, 1)))
    assert np.abs(x - np.array(t["x"])).max() < 1e-3
    t = g["poisson_2d"]
    sp = R.Space2((1, 8), (1, 7), library=lib)
    x = R.Poisson(sp, t["c"]).solve(np.tile(np.array(t["b_row"], float), (8, 1)))
    assert np.abs(x - np.array(t["x"])).max() < 1e-3
    # analytic round trips through the transforms (hholtz_adi.rs:248-308, poisson.rs:363-426)
    n = np.pi / 2
    for (k0, n0, fx, fac_h, fac_p) in (("cheb_dirichlet", 16, lambda x: np.cos(n * x), lambda a: 1 / (1 + a * n * n * 2), -1 / (n * n * 2)),
                                      ("fourier_r2c", 16, lambda x: np.cos(x), lambda a: 1 / (1 + a * n * n + a), -1 / (1 + n * n))):
        sp, osp = spaces(lib, k0, n0, "cheb_dirichlet", 7)
        x, y = osp.coords()
        v = fx(x)[:, None] * np.cos(n * y)[None, :]
        alpha = 1e-5
        out = sp.backward(R.HholtzAdi(sp, [alpha, alpha]).solve(sp.to_ortho(sp.forward(v))))
        assert np.abs(out - fac_h(alpha) * v).max() < 1e-3
        out = sp.backward(R.Poisson(sp, [1.0, 1.0]).solve(sp.to_ortho(sp.forward(v))))
        assert np.abs(out - fac_p * v).max() < 1e-3


def make_pair(lib, periodic, nx, ny, ra, pr, dt, aspect, eig_mode="parity", bc="rbc"):
    """Engine + oracle with the same deterministic initial condition.

    eig_mode "shared": the oracle runs the reference's algorithm on the ENGINE's x
    eigen-decomposition (rpde_navier2d_poisson_eigenbasis -> eig_override).  dgeev is setup, not
    part of the time step, and the Poisson solve with its 1e-10 eigenvalue shift
    (poisson.rs:84-87) amplifies the round-off of dgeev itself: two LAPACK runs on the same matrix
    (full vs per-parity blocks, or the same blocks assembled in a different order) differ by 1e-9
    in u and 5e-6 in p at 1025^2 and about ten times more per doubling of n, while a 4e-16
    perturbation of the initial condition changes the fields by 1e-15...1e-12 (measured with the
    oracle alone).  Step parity at the large sizes is therefore checked on a shared
    decomposition; the decomposition itself is checked by check_eigenbasis_is_valid."""
    ctor = "new_periodic" if periodic else "new_confined"
    nav = getattr(R.Navier2D, ctor)(nx, ny, ra, pr, dt, aspect, bc, library=lib)
    kw = {"eig_mode": eig_mode}
    if eig_mode == "shared":
        kw = {"eig_mode": "parity"} if periodic else {"eig_override": nav.poisson_eigenbasis()}
    ora = getattr(N.Navier2D, ctor)(nx, ny, ra, pr, dt, aspect, bc, **kw)
    for z in (nav, ora):
        z.set_velocity(0.2, 1.0, 1.0)
        z.set_temperature(0.2, 1.0, 1.0)
    return nav, ora


def check_step_parity(lib, periodic, nx, ny, ra, dt, steps, aspect=1.0, tol=1e-10, check_at=None, pr=1.0,
                      eig_mode="parity", bc="rbc"):
    """u, v, T, p (physical) after `steps` x update() vs the oracle; BASELINE.json bar 1e-10."""
    nav, ora = make_pair(lib, periodic, nx, ny, ra, pr, dt, aspect, eig_mode=eig_mode, bc=bc)
    for k in ("velx", "vely", "temp"):
        assert rel(getattr(nav, k).vhat, getattr(ora, k).vhat) < 1e-12, k
    check_at = set(check_at or [steps])
    worst = {}
    for s in range(1, steps + 1):
        nav.update()
        ora.update()
        if s in check_at:
            nf, of = nav.physical_fields(), ora.physical_fields()
            for k in of:
                e = rel(nf[k], of[k])
                worst[k] = max(worst.get(k, 0.0), e)
                assert e < tol, (k, s, e)
    assert abs(nav.get_time() - ora.time) < 1e-12
    assert abs(nav.div_norm() - ora.div_norm()) < 1e-9 * max(1.0, ora.div_norm())
    assert nav.exit() is False
    # callback diagnostics (functions.rs:146-233)
    for got, want in zip(nav.diagnostics(), (ora.eval_nu(), ora.eval_nuvol(), ora.eval_re())):
        assert abs(got - want) < 1e-9 * max(1.0, abs(want)), (got, want)
    return worst


def check_statistics(lib, periodic, nx, ny, ra=1e4, dt=0.01, tol=1e-11):
    """Statistics (statistics.rs:84-108, 248-271) on the device vs the oracle's restatement: three updates at
    different times; t_avg is a running mean, ux / uy / nusselt hold the last snapshot."""
    from oracle import navier as N
    nav, ora = make_pair(lib, periodic, nx, ny, ra, 1.0, dt, 1.0)
    st = R.Statistics.new(nav, 0.02, 0.04)
    so = N.Statistics(ora, 0.02, 0.04)
    nav.statistics = st
    assert st.num_save == 0 and st.avg_time == 0.0 and st.tot_time == nav.get_time()
    for k, steps in enumerate((2, 3, 1)):
        nav.update(steps)
        for _ in range(steps):
            ora.update()
        st.update()
        so.update_from(ora)
        assert st.num_save == so.num_save == k + 1
        assert abs(st.avg_time - so.avg_time) < 1e-12 and abs(st.tot_time - so.tot_time) < 1e-12
        for got, want in ((st.t_avg, so.t_avg), (st.ux_avg, so.ux_avg), (st.uy_avg, so.uy_avg), (st.nusselt, so.nusselt)):
            assert rel(got.vhat, want.vhat) < tol, (k, rel(got.vhat, want.vhat))
    # the mean is a mean: after three saves t_avg differs from the last snapshot
    assert rel(st.t_avg.vhat, ora.temp.to_ortho()) > 1e-6
    return st, so


def check_dct_line_backward(lib, n, nlines=5):
    """The whole-line transform kernel (csrc/dct_line.h) against the oracle: `backward` along an axis for the three
    Chebyshev bases, and backward_ortho(scale * d/dx to_ortho(.)) (the physical x-derivative S1 of the step needs)."""
    rng = np.random.default_rng(n)
    ortho = B.chebyshev(n)
    for kind, base in ((1, B.cheb_dirichlet(n)), (0, B.chebyshev(n)), (2, B.cheb_neumann(n))):
        m = n if kind == 0 else n - 2
        a = np.ascontiguousarray(rng.standard_normal((nlines, m)))
        out = np.empty((nlines, n))
        lib.call("rpde_dct_line_backward", kind, n, R._capi.ptr(a), nlines, R._capi.ptr(out), 0)
        want = base.backward(a, 1)
        assert rel(out, want) < 2e-12, (kind, n, rel(out, want))
        lib.call("rpde_dct_line_gradient", kind, n, R._capi.ptr(a), nlines, 0.5, R._capi.ptr(out), 0)
        want = ortho.backward_ortho(0.5 * ortho.differentiate(base.to_ortho(a, 1), 1, 1), 1)
        assert rel(out, want) < 2e-12, ("gradient", kind, n, rel(out, want))
    v = np.ascontiguousarray(rng.standard_normal((nlines, n)))
    for cut in (-1, 2 * n // 3):
        lib.call("rpde_dct_line_forward", n, R._capi.ptr(v), nlines, cut, R._capi.ptr(out), 0)
        want = ortho.forward_ortho(v, 1)
        if cut >= 0:
            want[:, cut:] = 0.0
        assert rel(out, want) < 2e-12, ("forward", n, cut, rel(out, want))
    lib.call("rpde_dct_line_backward", 0, n, R._capi.ptr(out), nlines, R._capi.ptr(v), 0)   # forward then backward of a dealiased line
    assert rel(ortho.forward_ortho(v, 1)[:, :2 * n // 3], want[:, :2 * n // 3]) < 1e-11


def check_conv_line(lib, n, nlines=5, lift=True):
    """The whole convection term per y-line (csrc/dct_line.h conv_line) against the oracle's operators: conv_term
    (functions.rs:56-69) summed and dealiased as in navier_eq.rs:56-101 -- backward of the x-derivative, backward of the
    y-derivative (to_ortho, Chebyshev recurrence), physical products, forward transform, 2/3 rule."""
    rng = np.random.default_rng(7 * n + nlines)
    ortho, dirichlet = B.chebyshev(n), B.cheb_dirichlet(n)
    decay = 1.0 / (1.0 + 1e-3 * np.arange(n - 2))
    fx = np.ascontiguousarray(rng.standard_normal((nlines, n - 2)) * decay)
    f0 = np.ascontiguousarray(rng.standard_normal((nlines, n - 2)) * decay)
    up, vp, bx, by = (np.ascontiguousarray(rng.standard_normal((nlines, n))) for _ in range(4))
    dscale, cut = 0.5, 2 * n // 3
    out = np.empty((nlines, n))
    null = R._capi.ptr(bx).__class__()        # NULL double*
    lib.call("rpde_conv_line", n, R._capi.ptr(fx), R._capi.ptr(f0), R._capi.ptr(up), R._capi.ptr(vp),
             R._capi.ptr(bx) if lift else null, R._capi.ptr(by) if lift else null, nlines, dscale, cut, R._capi.ptr(out), 0)
    a = dirichlet.backward(fx, 1)
    b = ortho.backward_ortho(dscale * ortho.differentiate(dirichlet.to_ortho(f0, 1), 1, 1), 1)
    if lift:
        a, b = a + bx, b + by
    want = ortho.forward_ortho(up * a + vp * b, 1)
    want[:, cut:] = 0.0
    e = rel(out, want)
    assert e < 2e-12, (n, lift, e)
    return e


def compare_with_independent_golden(nav, path, max_steps=None):
    """Engine (its OWN eigen-decomposition: one dgeev per parity block in C++) against a golden file written by
    tests/golden/make_headline_golden.py: the oracle in the REFERENCE's setup (eig_mode="full": one dgeev of the whole
    x operator, src/solver/utils.rs:67-99, fdma_tensor.rs:106-154), sub-sampled in physical space.  Returns
    {step: {field: (relative L2 over the sample points, the oracle's own full-vs-parity difference at that step)}}.
    `nav` must be freshly constructed with the golden's parameters; the deterministic IC is applied here."""
    g = np.load(path)
    stride = int(g["stride"])
    nav.set_velocity(0.2, 1.0, 1.0)
    nav.set_temperature(0.2, 1.0, 1.0)
    out, done = {}, 0
    for s in [int(v) for v in g["snaps"]]:
        if f"velx_{s}" not in g.files or (max_steps is not None and s > max_steps):
            continue
        nav.update(s - done)
        done = s
        f = nav.physical_fields()
        out[s] = {}
        for k in ("velx", "vely", "temp", "pres"):
            want = g[f"{k}_{s}"]
            got = f[k][::stride, ::stride]
            out[s][k] = (float(np.linalg.norm(got - want) / np.linalg.norm(want)), float(g[f"{k}_{s}_full_vs_parity"]))
            # the full-field norm of the golden run pins the points between the 65 x 65 samples as well
            gn = float(g[f"{k}_{s}_norm"])
            out[s][k + "_norm"] = (abs(float(np.linalg.norm(f[k])) - gn) / gn, float("nan"))
    return out


from tests.bounds import independent_golden_bound   # noqa: E402,F401  (one definition, shared with bench.py)


def check_config2_golden(lib):
    """BASELINE.json configs[1] against the committed oracle samples (tests/golden/make_config2_golden.py: the oracle in
    the REFERENCE's setup, one dgeev of the whole operator -- independent of the engine's); bounds:
    independent_golden_bound, i.e. 1e-10 for u, v, T, p after 100 and 200 steps."""
    g = np.load(os.path.join(GOLDEN, "config2_1025_200steps.npz"))
    assert str(g["eig_mode"]) == "full", "config-2 golden must come from the reference's one-dgeev setup"
    nx, ny, stride = int(g["nx"]), int(g["ny"]), int(g["stride"])
    nav = R.Navier2D.new_confined(nx, ny, float(g["ra"]), float(g["pr"]), float(g["dt"]), 1.0, "rbc", library=lib)
    nav.set_velocity(0.2, 1.0, 1.0)
    nav.set_temperature(0.2, 1.0, 1.0)
    done, report = 0, {}
    for s in (10, 100, 200):
        nav.update(s - done)
        done = s
        f = nav.physical_fields()
        for k in ("velx", "vely", "temp", "pres"):
            want = g[f"{k}_{s}"]
            got = f[k][::stride, ::stride]
            tol = independent_golden_bound(float(g[f"{k}_{s}_full_vs_parity"]), n=nx)   # (own pair x the spread ratio measured at this size)
            if s >= 100:
                assert tol == 1e-10, (k, s, tol)     # the golden itself must have left the transient: the 1e-10 bar is exercised
            err = np.linalg.norm(got - want) / np.linalg.norm(want)
            report[(k, s)] = float(err)
            assert err < tol, (k, s, err, tol)
            # the full-field norm pins the points between the samples as well
            assert abs(np.linalg.norm(f[k]) - float(g[f"{k}_{s}_norm"])) < tol * float(g[f"{k}_{s}_norm"]), (k, s)
    print({f"{k}@{s}": f"{e:.1e}" for (k, s), e in report.items()})
    assert abs(nav.div_norm() - float(g["div_norm"])) < 1e-8 * max(1.0, float(g["div_norm"]))


def check_independent_golden(lib, n):
    """Engine (own setup) against the golden of the oracle in the reference's setup, n x n; bounds: independent_golden_bound."""
    path = os.path.join(GOLDEN, f"headline_{n}_full.npz")
    g = np.load(path)
    nav = R.Navier2D.new_confined(n, n, float(g["ra"]), float(g["pr"]), float(g["dt"]), 1.0, "rbc", library=lib)
    res = compare_with_independent_golden(nav, path)
    assert len(res) >= 9, f"golden {path} holds {len(res)} snapshots, expected at least 9"
    print({s: {k: f"{e:.1e} (oracle full vs parity {b:.1e})" for k, (e, b) in r.items() if not k.endswith("_norm")} for s, r in res.items()})
    for s, r in res.items():
        for k, (err, fvp) in r.items():
            if k.endswith("_norm"):      # |norm(engine) - norm(golden)| / norm(golden) <= the relative L2 bound of that field
                assert err < independent_golden_bound(r[k[:-5]][1], n=n, step=s, field=k[:-5]), (n, s, k, err)
                continue
            bound = independent_golden_bound(fvp, n=n, step=s, field=k)
            assert err < bound, (n, s, k, err, bound, fvp)


def check_shared_basis_golden(lib, path, tol=1e-10, max_steps=None):
    """The SAME-INPUTS comparison over the full horizon (tests/golden/make_shared_basis_golden.py): the engine is created with
    the x eigenvalues the golden file carries (`x_spectrum=`: the library rebuilds the oracle's eigenbasis bit for bit, no
    LAPACK), runs the golden's workload, and EVERY snapshot is held to the plain `tol` (BASELINE.json: 1e-10) on u, v, T and p
    -- sample points and full-field norms.  Returns {step: {field: rel L2}}."""
    g = np.load(path)
    n, stride = int(g["nx"]), int(g["stride"])
    lam_in = np.ascontiguousarray(g["x_spectrum"])
    # the reproducibility the golden rests on: the same refined eigenvalues from the same input, bit for bit
    # (host arithmetic only: a few seconds at 4097 -- checked at every size since round 6)
    ref, _, _ = R.poisson_x_eigenbasis_from_spectrum((R.CHEB_NEUMANN, n), 1.0, lam_in, library=lib)
    if lib.is_device_build and str(g["library_version"]) == lib.version:
        assert np.array_equal(ref, g["x_spectrum_refined"]), "the library does not reproduce the golden's eigenvalues bit for bit"
    nav = R.Navier2D.new_confined(n, int(g["ny"]), float(g["ra"]), float(g["pr"]), float(g["dt"]), 1.0, "rbc", library=lib,
                                  init_random=None, x_spectrum=lam_in)
    nav.set_velocity(0.2, 1.0, 1.0)
    nav.set_temperature(0.2, 1.0, 1.0)
    out, done = {}, 0
    for s in [int(v) for v in g["snaps"]]:
        if f"velx_{s}" not in g.files or (max_steps is not None and s > max_steps):
            continue
        nav.update(s - done)
        done = s
        f = nav.physical_fields()
        out[s] = {}
        for k in ("velx", "vely", "temp", "pres"):
            want = g[f"{k}_{s}"]
            e = float(np.linalg.norm(f[k][::stride, ::stride] - want) / np.linalg.norm(want))
            out[s][k] = e
            assert e < tol, (n, s, k, e)
            assert abs(np.linalg.norm(f[k]) - float(g[f"{k}_{s}_norm"])) < tol * float(g[f"{k}_{s}_norm"]), (n, s, k, "norm")
        assert abs(nav.div_norm() - float(g[f"div_norm_{s}"])) < 1e-8 * max(1.0, float(g[f"div_norm_{s}"]))
        print(s, {k: f"{e:.1e}" for k, e in out[s].items()})
    assert out, "no snapshot compared"
    return out


def check_extended_golden(lib, n=4097):
    """The engine (own setup) against the EXTENDED golden tests/golden/headline_<n>_full_extended.npz: a second run of the
    oracle in the reference's one-dgeev setup, carried on to step 800 (make_headline_golden.py, RPDE_GOLDEN_SNAPS).  The bar
    is the PLAIN 1e-10 of BASELINE.json on u, v, T and p -- no envelope: there must be a snapshot from which ALL four fields
    are below 1e-10 and stay below it at every later snapshot, and u, v, T must be below it from step 200 on.  Before
    that snapshot the pressure carries the start-up transient of two independent dgeev runs (two runs of
    the reference's own setup are 1.7e-9 apart in p at step 200).  Returns (rows, first step with all fields < 1e-10)."""
    path = os.path.join(GOLDEN, f"headline_{n}_full_extended.npz")
    g = np.load(path)
    nav = R.Navier2D.new_confined(n, n, float(g["ra"]), float(g["pr"]), float(g["dt"]), 1.0, "rbc", library=lib)
    res = compare_with_independent_golden(nav, path)
    norms = {s: {k: e for k, (e, _) in r.items() if k.endswith("_norm")} for s, r in res.items()}
    rows = {s: {k: e for k, (e, _) in r.items() if not k.endswith("_norm")} for s, r in res.items()}
    for s, r in rows.items():
        print(s, {k: f"{e:.2e}" for k, e in r.items()})
    assert max(rows) >= 800, f"extended golden ends at step {max(rows)}"
    below = [s for s in sorted(rows) if all(e < 1e-10 for e in rows[s].values())]
    first = None
    for s in sorted(rows, reverse=True):          # the last run of consecutive snapshots below the bar
        if all(e < 1e-10 for e in rows[s].values()):
            first = s
        else:
            break
    assert first is not None, ("no snapshot with u, v, T, p all below 1e-10", rows[max(rows)])
    assert below and below[0] == first, ("fields rise above 1e-10 again after a snapshot below it", below, first)
    for s in rows:
        if s >= 200:
            for k in ("velx", "vely", "temp"):
                assert rows[s][k] < 1e-10, (s, k, rows[s][k])
    for s in rows:            # full-field norms: wherever a field is below the bar at its samples, its norm is too
        for k, e in rows[s].items():
            if e < 1e-10:
                assert norms[s][k + "_norm"] < 1e-10, (s, k, norms[s][k + "_norm"])
    print("first snapshot with u, v, T, p all below 1e-10:", first)
    return rows, first


def check_ab_switch(lib, switch, nx, ny, steps, tol=1e-11, off_value="0"):
    """The confined step with an A/B switch of the engine on (default) and off (<switch>=0, or `off_value` for a switch whose
    default is off): same engine, same setup data."""
    import rustpde_mpi_amd as R
    fields, kinds = {}, {}
    for flag in ("1", "0"):
        if flag == "1":
            os.environ.pop(switch, None)
        else:
            os.environ[switch] = off_value
        nav = R.Navier2D.new_confined(nx, ny, 1e7, 1.0, 1e-3, 1.0, "rbc", library=lib, init_random=None)
        nav.set_velocity(0.2, 1.0, 1.0); nav.set_temperature(0.2, 1.0, 1.0)
        nav.update(steps)
        fields[flag] = nav.physical_fields()
        kinds[flag] = sorted({kind for _, _, _, _, kind in nav.schedule() if kind.startswith("whole-line")})
        del nav
    os.environ.pop(switch, None)
    assert kinds["1"], kinds
    if switch == "RPDE_WHOLE_LINE":
        assert not kinds["0"], kinds
    for k in fields["0"]:
        e = rel(fields["1"][k], fields["0"][k])
        assert e < tol, (switch, k, e)
    print(switch, kinds)


def run_isolated(call, timeout=900):
    """Run `checks.<call>` (an expression like "check_config2_golden(lib)", with lib = the product library) in a child
    process: a large engine gets a fresh HIP context and its HBM back at exit, and a device fault -- which ends the process
    that owns the GPU context (the HIP runtime aborts) -- fails THIS test instead of ending the pytest session.  There is no
    second run: any failure of the child, device fault included, fails the test with the child's output."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import rustpde_mpi_amd as R\nfrom tests import checks as K\nlib = R.lib()\nassert lib.is_device_build\n"
            f"K.{call}\nprint('ISOLATED-OK')\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=timeout)
    ok = r.returncode == 0 and "ISOLATED-OK" in r.stdout
    sys.stdout.write(r.stdout[-4000:])
    assert ok, f"child process ended with code {r.returncode}\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}"
