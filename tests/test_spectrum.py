"""Setup data supplied by the host: the Poisson solver's x eigenvalues (rpde_poisson_x_spectrum,
rpde_poisson_x_eigenbasis_from_spectrum, rpde_navier2d_create_confined_with_spectrum, rpde_poisson_create_with_spectrum).

What the mechanism is for: the reference's setup (LAPACK dgeev, src/solver/utils.rs:67-99) is not reproducible across thread
counts or CPU models and Poisson::new's -1e-10 shift amplifies the difference by 1e10; with the
eigenvalues as data and the vectors rebuilt WITHOUT LAPACK in a fixed operation order, a checker anywhere runs the reference
algorithm on exactly the engine's setup data -- the same-inputs goldens tests/golden/shared_basis_*.npz."""
import os

import numpy as np
import pytest

import rustpde_mpi_amd as R
from oracle import bases as B, navier as N, solver as S
from tests import checks as K


def _pencil(n):
    (_, a), (_, b), _ = S.ingredients_for_hholtz(B.cheb_neumann(n))
    bd, bu = b
    z = np.zeros_like(bd)
    return S.band_to_dense(z, bd, bu, z), S.band_to_dense(*a)     # A = laplacian (c0 = 1), C = mass


def check_spectrum_basis(lib, n):
    """Eigenpairs of the banded pencil to round-off (far better than dgeev's on the dense inv(C) A), descending order kept,
    biorthogonality fwd C bwd = I, and bit-identical output on a second call."""
    lam_in = R.poisson_x_spectrum((R.CHEB_NEUMANN, n), 1.0, library=lib)
    lam, fwd, bwd = R.poisson_x_eigenbasis_from_spectrum((R.CHEB_NEUMANN, n), 1.0, lam_in, library=lib)
    m, me = n - 2, (n - 1) // 2
    assert np.all(np.diff(lam[:me]) < 0) and np.all(np.diff(lam[me:]) < 0), "each parity block descending"
    assert abs(lam[0]) < 1e-12 and lam[1:].max() < 0, "one zero mode (Neumann), the rest negative"
    assert np.abs((lam - lam_in) / np.maximum(np.abs(lam_in), 1.0)).max() < 1e-4, "refinement stays at its eigenvalue"
    A, C = _pencil(n)
    AQ, CQ = A @ bwd, C @ bwd
    res = np.linalg.norm(AQ - CQ * lam[None, :], axis=0) / (np.linalg.norm(AQ, axis=0) + np.abs(lam) * np.linalg.norm(CQ, axis=0) + 1e-300)
    res = res[np.abs(lam) > 1e-9]      # (the zero mode: A q = 0 = lam C q, no scale to compare with)
    assert np.median(res) < 1e-12 and res.max() < 1e-7, (np.median(res), np.sort(res)[-3:])
    ident = np.abs(fwd @ C @ bwd - np.eye(m)).max()
    assert ident < 1e-4, ident        # the two or three largest (spurious, O(n^4)) modes carry all of it
    lam2, fwd2, bwd2 = R.poisson_x_eigenbasis_from_spectrum((R.CHEB_NEUMANN, n), 1.0, lam_in, library=lib)
    assert np.array_equal(lam, lam2) and np.array_equal(fwd, fwd2) and np.array_equal(bwd, bwd2)
    return lam_in, lam, fwd, bwd


@pytest.mark.parametrize("n", [18, 65, 257])
def test_product_library_spectrum_basis_host_only(n):
    """The host-only entry points of the PRODUCT library (no GPU needed)."""
    if not os.path.exists(R.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    check_spectrum_basis(R.lib(), n)


def test_product_library_reproduces_the_goldens_spectrum_bit_for_bit():
    """The claim the same-inputs goldens rest on: from the committed input eigenvalues the library rebuilds the committed
    refined eigenvalues exactly (this box, the driver's box, the GPU box: the same binary)."""
    lib, seen = R.lib(), 0
    for f in sorted(os.listdir(K.GOLDEN)):
        if not (f.startswith("shared_basis_") and f.endswith(".npz")):
            continue
        g = np.load(os.path.join(K.GOLDEN, f))
        if str(g["library_version"]) != lib.version:
            continue
        n = int(g["nx"])
        lam, _, _ = R.poisson_x_eigenbasis_from_spectrum((R.CHEB_NEUMANN, n), 1.0, np.ascontiguousarray(g["x_spectrum"]), library=lib)
        assert np.array_equal(lam, g["x_spectrum_refined"]), f
        seen += 1
    if not seen:
        pytest.skip("no shared_basis_*.npz committed yet")


@pytest.mark.parametrize("n,ra,dt,steps", [(65, 1e6, 1e-3, 20), (129, 1e7, 1e-3, 8)])
def test_emu_engine_on_a_supplied_spectrum_equals_oracle_on_the_same_basis(emu_lib, n, ra, dt, steps):
    """Engine created with x_spectrum: its eigenbasis IS the host function's (bit for bit), and the oracle on that basis
    agrees to 1e-10 in u, v, T and p from the FIRST step (no start-up transient: same inputs)."""
    lam_in, lam, fwd, bwd = check_spectrum_basis(emu_lib, n)
    nav = R.Navier2D.new_confined(n, n, ra, 1.0, dt, 1.0, "rbc", library=emu_lib, init_random=None, x_spectrum=lam_in)
    l2, f2, b2 = nav.poisson_eigenbasis()
    assert np.array_equal(l2, lam) and np.array_equal(f2, fwd) and np.array_equal(b2, bwd)
    ora = N.Navier2D.new_confined(n, n, ra, 1.0, dt, 1.0, "rbc", eig_override=(lam, fwd, bwd))
    for z in (nav, ora):
        z.set_velocity(0.2, 1.0, 1.0)
        z.set_temperature(0.2, 1.0, 1.0)
    for s in range(steps):
        nav.update(1)
        ora.update()
        got, want = nav.physical_fields(), ora.physical_fields()
        for k in want:
            assert K.rel(got[k], want[k]) < 1e-10, (s, k, K.rel(got[k], want[k]))


def test_emu_poisson_operator_with_spectrum(emu_lib):
    n0, n1 = 65, 33
    lam_in = R.poisson_x_spectrum((R.CHEB_NEUMANN, n0), 1.0, library=emu_lib)
    lam, fwd, bwd = R.poisson_x_eigenbasis_from_spectrum((R.CHEB_NEUMANN, n0), 1.0, lam_in, library=emu_lib)
    sp = R.Space2((R.CHEB_NEUMANN, n0), (R.CHEB_NEUMANN, n1), library=emu_lib)
    ps = R.Poisson(sp, [1.0, 1.0], x_spectrum=lam_in)
    l2, f2, b2 = ps.eigenbasis()
    assert np.array_equal(l2, lam) and np.array_equal(f2, fwd) and np.array_equal(b2, bwd)
    osp = B.Space2(B.cheb_neumann(n0), B.cheb_neumann(n1))
    f = N.Field2(osp)
    f.v = np.random.default_rng(2).standard_normal(f.v.shape)
    f.forward()
    rhs = f.to_ortho()
    want = S.Poisson(osp, [1.0, 1.0], eig_override=(lam, fwd, bwd)).solve(rhs)
    got = ps.solve(rhs)
    want[0, 0] = got[0, 0] = 0.0
    assert K.rel(got, want) < 1e-11
    own = S.Poisson(osp, [1.0, 1.0], eig_mode="parity").solve(rhs)      # and against the oracle's own LAPACK basis
    own[0, 0] = 0.0
    assert K.rel(got, own) < 1e-8


def test_emu_spectrum_errors(emu_lib):
    lam = R.poisson_x_spectrum((R.CHEB_NEUMANN, 33), 1.0, library=emu_lib)
    with pytest.raises(R.RpdeError, match="nx - 2 eigenvalues"):
        R.Navier2D.new_confined(35, 33, 1e5, 1.0, 0.01, 1.0, "rbc", library=emu_lib, init_random=None, x_spectrum=lam)
    with pytest.raises(R.RpdeError, match="confined, one device"):
        R.Navier2D._new("rpde_navier2d_create_periodic", 32, 33, 1e5, 1.0, 0.01, 1.0, "rbc", 0, emu_lib, True, None, lam)
    with pytest.raises(R.RpdeError, match="two-term stencil"):
        R.poisson_x_spectrum((R.CHEBYSHEV, 33), 1.0, library=emu_lib)
    # an engine built right after one with a spectrum does not inherit it
    a = R.Navier2D.new_confined(33, 33, 1e5, 1.0, 0.01, 1.0, "rbc", library=emu_lib, init_random=None, x_spectrum=lam)
    b = R.Navier2D.new_confined(33, 33, 1e5, 1.0, 0.01, 1.0, "rbc", library=emu_lib, init_random=None)
    assert not np.array_equal(a.poisson_eigenbasis()[2], b.poisson_eigenbasis()[2])
