#!/bin/bash
# round 3, call 11: where the 1025^2 step stands (r02: 0.738 ms): per-launch table, graph replay vs plain launches
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03k
rm -rf $O; mkdir -p $O
timeout 200 python tools/profile_step.py 1025 1025 > $O/profile_1025.txt 2>&1; cut -c1-100 $O/profile_1025.txt
RPDE_GEMM_SWIZZLE=0 timeout 200 python tools/profile_step.py 1025 1025 > $O/profile_1025_noswz.txt 2>&1; grep -E "^G1|^G2|^total" $O/profile_1025_noswz.txt | cut -c1-100
python - <<'PY' 2>&1 | tee $O/graph_1025.txt
import time, rustpde_mpi_amd as R
for n in (1025, 513):
    nav = R.Navier2D.new_confined(n, n, 1e7, 1.0, 1e-3, 1.0, "rbc")
    nav.set_velocity(0.2, 1.0, 1.0); nav.set_temperature(0.2, 1.0, 1.0)
    nav.update(20)
    t0 = time.time(); nav.update(400); dt = time.time() - t0
    print(f"{n}^2 graph replay: {dt/400*1e3:.4f} ms/step wall, device {nav.last_update_ms/400:.4f} ms/step")
PY
