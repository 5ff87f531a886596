"""CPU oracle for the rustpde Navier2D hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, in NumPy/SciPy, the algorithm of the reference's
``Navier2D::update()`` (``src/navier_stokes/navier.rs:438-466`` of
preiter93/rustpde-mpi) together with the pieces of the external crate
``funspace 0.3.0`` it needs (bases, transforms, stencils, differentiation,
solver matrices; the crate's source is NOT vendored in the reference tree, so
its published algorithm is restated from SURVEY.md App. A).

Rules (enforced by tests/test_layout.py):
  * Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
    of ``bench.py`` may import this package.  The product path
    (``rustpde_mpi_amd``) never imports it and has no CPU fallback.
  * It is the checker, never the thing measured or shipped.

Parity pin status
-----------------
* Solver level (``HholtzAdi``, ``Poisson`` and through them ``mass``,
  ``laplace_inv``, ``laplace_inv_eye``, the Dirichlet stencil, ``Fdma``,
  ``MatVecFdma``, the eigen-decomposition path): PINNED against the
  reference's own known-answer vectors (``src/solver/hholtz_adi.rs:192-246``,
  ``src/solver/poisson.rs:274-361``) and analytic round trips
  (``hholtz_adi.rs:248-308``, ``poisson.rs:363-426``); see
  ``tests/test_oracle_golden.py`` and ``tests/golden/reference_known_answers.json``.
* Step level (``update()``): **parity unpinned** -- the reference holds no
  golden output for a time step and cannot be built here (no Rust toolchain,
  funspace/hdf5/MPI absent).  The step-level oracle is defined by this
  restatement; its pieces are cross-checked by the pinned tests above, and the
  step as a whole is pinned PHYSICALLY: the critical Rayleigh number of the
  square no-slip cavity extracted from its growth rates (dt -> 0) is 2585.5
  against the literature value 2585.02, and the threshold of the periodic
  layer is 1708.2 against Chandrasekhar's 1707.762 (``tests/test_physics_pin.py``).
"""
from . import bases, solver, navier  # noqa: F401
