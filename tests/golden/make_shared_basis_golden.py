"""Generate tests/golden/shared_basis_<n>.npz: the SAME-INPUTS golden of the confined step over its full horizon.

The reference's Poisson solver diagonalises the x operator with LAPACK dgeev (src/solver/utils.rs:67-99,
fdma_tensor.rs:123-127) and shifts the spectrum by -1e-10 (poisson.rs:84-87): dgeev's output is not reproducible across
thread counts or CPU models, and the shift amplifies the difference by 1e10 -- two runs of the reference's own setup differ
by 1.7e-9 in p after 200 steps at 4097^2 (headline_4097_two_reference_setups.json).  A comparison of the TIME STEP over 200
steps therefore needs both sides on the same setup data.  This script makes that possible across machines:

  1. `rpde_poisson_x_spectrum` (host only, in the product library): the x eigenvalues, LAPACK dgeev, values only;
  2. `rpde_poisson_x_eigenbasis_from_spectrum` (host only, no LAPACK, bit-reproducible): refined eigenvalues, fwd, bwd;
  3. the CPU oracle (oracle/navier.py, the restatement of Navier2D::update) runs the workload on that eigenbasis
     (`eig_override`) and its sub-sampled fields are committed TOGETHER WITH THE EIGENVALUES of step 1;
  4. on the GPU box the engine is created with `x_spectrum=` those eigenvalues: the same library binary rebuilds the same
     eigenbasis bit for bit (tests/test_gpu_parity.py::test_shared_basis_golden, bench.py `parity_shared_basis_golden`),
     and every snapshot is held to the plain 1e-10 of BASELINE.json on u, v, T and p.

    python tests/golden/make_shared_basis_golden.py 4097          (about 3 h on 8 cores; n = 1025: 5 min)
    python tests/golden/make_shared_basis_golden.py 1025 1e7 1e-3  (BASELINE config 2's parameters)

RPDE_GOLDEN_SNAPS="1,2,4" limits the snapshots; partial results are written after every snapshot."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

SNAPS = (1, 2, 4, 10, 20, 50, 100, 150, 200)
if os.environ.get("RPDE_GOLDEN_SNAPS"):
    SNAPS = tuple(int(v) for v in os.environ["RPDE_GOLDEN_SNAPS"].split(","))
FIELDS = ("velx", "vely", "temp", "pres")


def main():
    import rustpde_mpi_amd as R
    from oracle import navier as N
    n = int(sys.argv[1])
    ra = float(sys.argv[2]) if len(sys.argv) > 2 else 1e8
    dt = float(sys.argv[3]) if len(sys.argv) > 3 else 2e-4
    tag = "" if len(sys.argv) <= 2 else f"_ra{ra:g}_dt{dt:g}"
    out_path = os.path.join(ROOT, "tests", "golden", f"shared_basis_{n}{tag}.npz")
    lib = R.lib()           # the PRODUCT library: its host-only setup code is what the engine on the GPU box runs
    assert lib.is_device_build
    stride = max(1, (n - 1) // 64)
    t0 = time.time()
    lam_in = R.poisson_x_spectrum((R.CHEB_NEUMANN, n), 1.0, library=lib)
    lam, fwd, bwd = R.poisson_x_eigenbasis_from_spectrum((R.CHEB_NEUMANN, n), 1.0, lam_in, library=lib)
    print(f"spectrum + eigenbasis {time.time() - t0:.1f} s", flush=True)
    nav = N.Navier2D.new_confined(n, n, ra, 1.0, dt, 1.0, "rbc", eig_override=(lam, fwd, bwd))
    del fwd, bwd
    nav.set_velocity(0.2, 1.0, 1.0)
    nav.set_temperature(0.2, 1.0, 1.0)
    out = dict(nx=n, ny=n, ra=ra, pr=1.0, dt=dt, stride=stride, x_spectrum=lam_in, x_spectrum_refined=lam,
               library_version=lib.version, snaps=np.array(SNAPS))
    t0 = time.time()
    for s in range(1, max(SNAPS) + 1):
        nav.update()
        if s in SNAPS:
            f = nav.physical_fields()
            for k in FIELDS:
                out[f"{k}_{s}"] = f[k][::stride, ::stride].copy()
                out[f"{k}_{s}_norm"] = np.array(np.linalg.norm(f[k]))
            out[f"div_norm_{s}"] = np.array(nav.div_norm())
            np.savez_compressed(out_path + ".tmp.npz", **out)
            os.replace(out_path + ".tmp.npz", out_path)
            print(s, f"{time.time() - t0:.0f} s", {k: float(np.abs(v).max()) for k, v in f.items()}, flush=True)
    print("wrote", out_path, os.path.getsize(out_path))


if __name__ == "__main__":
    main()
