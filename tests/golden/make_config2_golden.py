"""Generate tests/golden/config2_1025_200steps.npz: BASELINE.json configs[1]
(Navier2D::new_confined 1025 x 1025, Ra = 1e7, Pr = 1, aspect = 1, bc = rbc, deterministic IC of
examples/navier_rbc.rs) advanced 200 steps with the CPU oracle, sub-sampled every 16th grid point
(65 x 65 samples per field).  dt = 1e-3: the explicit convection needs dt of this order at n = 1025
(time-step caveat); the value is stored in the file.

    python tests/golden/make_config2_golden.py        (about 8 minutes on 8 cores: two oracle runs)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import navier as N  # noqa: E402

NX = NY = 1025
RA, PR, DT, STEPS, STRIDE = 1e7, 1.0, 1e-3, 200, 16


def main():
    # The golden is the oracle in the REFERENCE's setup: ONE dgeev of the whole x operator (eig_mode="full",
    # src/solver/utils.rs:67-99, fdma_tensor.rs:106-154) -- independent of the engine, which runs one dgeev per parity
    # block in its own C++ setup code (round 3 generated this file with eig_mode="parity", the engine's own algorithm).
    # A second oracle run with eig_mode="parity" measures how far two valid LAPACK eigenbases of the same matrix are
    # apart at each snapshot (`<field>_<step>_full_vs_parity`; tests/checks.py independent_golden_bound).
    nav = N.Navier2D.new_confined(NX, NY, RA, PR, DT, 1.0, "rbc", eig_mode="full")
    alt = N.Navier2D.new_confined(NX, NY, RA, PR, DT, 1.0, "rbc", eig_mode="parity")
    for z in (nav, alt):
        z.set_velocity(0.2, 1.0, 1.0)
        z.set_temperature(0.2, 1.0, 1.0)
    t0 = time.time()
    snaps = {}
    for s in range(1, STEPS + 1):
        nav.update()
        alt.update()
        if s in (10, 100, 200):
            f, fa = nav.physical_fields(), alt.physical_fields()
            for k, v in f.items():
                snaps[f"{k}_{s}"] = v[::STRIDE, ::STRIDE].copy()
                snaps[f"{k}_{s}_norm"] = np.array(np.linalg.norm(v))
                snaps[f"{k}_{s}_full_vs_parity"] = np.array(np.linalg.norm(v - fa[k]) / np.linalg.norm(v))
            print(s, time.time() - t0, {k: float(np.abs(v).max()) for k, v in f.items()},
                  {k: float(snaps[f"{k}_{s}_full_vs_parity"]) for k in f}, flush=True)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config2_1025_200steps.npz")
    np.savez_compressed(out, nx=NX, ny=NY, ra=RA, pr=PR, dt=DT, steps=STEPS, stride=STRIDE, eig_mode="full",
                        div_norm=nav.div_norm(), **snaps)
    print("wrote", out, os.path.getsize(out))


if __name__ == "__main__":
    main()
