#!/usr/bin/env python
"""Full per-launch table of one step (HIP events around every launch): tools/profile_step.py [nx ny]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustpde_mpi_amd as R
if os.environ.get("RPDE_TOOLS_LIB"):      # an experiment build (rustpde_mpi_amd.build variant): librustpde_hip_<variant>.so
    from rustpde_mpi_amd._capi import Lib
    R._lib = Lib(os.path.join(os.path.dirname(R.LIB_PATH), os.environ["RPDE_TOOLS_LIB"]))

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 4097
ny = int(sys.argv[2]) if len(sys.argv) > 2 else 4097
per = len(sys.argv) > 3 and sys.argv[3] == "periodic"
# RPDE_TOOLS_SPECTRUM=<file.npy>: the x eigenvalues are computed once (rpde_poisson_x_spectrum) and kept in that file; later
# runs create the engine from them (no dgeev: seconds instead of a minute at 4097^2) -- A/B runs of one gpurun call
spec = None
if os.environ.get("RPDE_TOOLS_SPECTRUM") and not per:
    import numpy as np
    path = os.environ["RPDE_TOOLS_SPECTRUM"]
    if not os.path.exists(path):
        np.save(path, R.poisson_x_spectrum((R.CHEB_NEUMANN, nx), 1.0))
    spec = np.load(path)
    assert spec.size == nx - 2
kw = {"x_spectrum": spec} if spec is not None else {}
nav = (R.Navier2D.new_periodic if per else R.Navier2D.new_confined)(nx, ny, 1e8, 1.0, 2e-4, 1.0, "rbc", **kw)
nav.set_velocity(0.2, 1.0, 1.0); nav.set_temperature(0.2, 1.0, 1.0)
nav.profile(2)
rows = nav.profile(5)
tot = sum(r["ms_total"] for r in rows) / 5
print(f"{'tag':36s} {'n/step':>6s} {'ms/launch':>10s} {'ms/step':>8s} {'share':>6s} {'GB/s':>8s} {'TF/s':>6s}")
for r in rows:
    n = r["launches"] / 5; ms = r["ms_total"] / r["launches"]
    print(f"{r['tag']:36s} {n:6.0f} {ms:10.4f} {ms*n:8.3f} {ms*n/tot:6.3f} {r['bytes']/ms/1e6:8.1f} {r['flops']/ms/1e9:6.2f}")
print(f"total {tot:.3f} ms/step")
