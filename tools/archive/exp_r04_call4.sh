#!/bin/bash
# round 4, call 4: single-pass column scans, second form (scalar coefficient loads, agent-scope relaxed atomics instead of
# fences, aggregates staged through LDS by all waves, chains in place) -- parity on the device, then per-launch times
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04d; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "one_pass or confined_257 or confined_step" 2>&1 | tail -8
for cfg in "w8:X=1" "w16:RPDE_COL1_W=16" "w4:RPDE_COL1_W=4" "off:RPDE_COL_ONEPASS=0"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs python tools/profile_step.py 4097 4097 > $O/p4097_$name.txt 2>&1
  echo "== 4097 $name"; grep "column scan\|total" $O/p4097_$name.txt
done
for cfg in "w8:X=1" "w4:RPDE_COL1_W=4" "w16:RPDE_COL1_W=16" "off:RPDE_COL_ONEPASS=0"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs python tools/profile_step.py 1025 1025 > $O/p1025_$name.txt 2>&1
  echo "== 1025 $name"; grep "column scan\|total" $O/p1025_$name.txt
done
for cfg in "w8:X=1" "off:RPDE_COL_ONEPASS=0"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs python tools/profile_step.py 2049 2049 > $O/p2049_$name.txt 2>&1
  echo "== 2049 $name"; grep "column scan\|total" $O/p2049_$name.txt
done
python bench.py --no-cpu-baseline --nx 1025 --ny 1025 --ra 1e7 --dt 1e-3 --steps 200 > $O/bench_1025.json 2>$O/bench_1025.err; tail -c 300 $O/bench_1025.json; tail -3 $O/bench_1025.err
