#!/bin/bash
# round 4, call 3: single-pass column scans (colscan1.h) -- parity on the device, then per-launch times against the three kernels
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04c; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "one_pass or confined_257 or confined_step" 2>&1 | tail -8
for cfg in "auto:X=1" "w8:RPDE_COL1_W=8" "w4:RPDE_COL1_W=4" "off:RPDE_COL_ONEPASS=0"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs python tools/profile_step.py 4097 4097 > $O/p4097_$name.txt 2>&1
  echo "== 4097 $name"; grep "column scan\|total" $O/p4097_$name.txt
done
for cfg in "auto:X=1" "w8:RPDE_COL1_W=8" "w16:RPDE_COL1_W=16" "off:RPDE_COL_ONEPASS=0"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs python tools/profile_step.py 1025 1025 > $O/p1025_$name.txt 2>&1
  echo "== 1025 $name"; grep "column scan\|total" $O/p1025_$name.txt
done
python bench.py --no-cpu-baseline --nx 1025 --ny 1025 --ra 1e7 --dt 1e-3 --steps 200 > $O/bench_1025.json 2>$O/bench_1025.err; tail -c 600 $O/bench_1025.json
