#!/bin/bash
# round 4, call 5: (a) column scans, the tile's LAST workgroup turns aggregates into inflow states (O(NSB) exchange): W = 8 vs 16;
# (b) S1 with the state line loaded once (hdct_pair_line: two halves of a workgroup, one transform each) vs the two transforms in a row
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04e; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "one_pass or confined_257 or confined_step or (whole_line_stage and S1) or step_through_the_whole or space_ops_4097" 2>&1 | tail -8
for cfg in "w8:X=1" "w16:RPDE_COL1_W=16" "nopair:RPDE_S1_PAIR=0" "w16b:RPDE_COL1_W=16"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs python tools/profile_step.py 4097 4097 > $O/p4097_$name.txt 2>&1
  echo "== 4097 $name"; grep "column scan\|S1 x\|total" $O/p4097_$name.txt
done
for cfg in "w8:X=1" "w16:RPDE_COL1_W=16" "nopair:RPDE_S1_PAIR=0"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs python tools/profile_step.py 1025 1025 > $O/p1025_$name.txt 2>&1
  echo "== 1025 $name"; grep "column scan\|S1 x\|total" $O/p1025_$name.txt
done
for cfg in "w8:X=1" "w16:RPDE_COL1_W=16"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs python tools/profile_step.py 2049 2049 > $O/p2049_$name.txt 2>&1
  echo "== 2049 $name"; grep "column scan\|S1 x\|total" $O/p2049_$name.txt
done
python bench.py --no-cpu-baseline --nx 1025 --ny 1025 --ra 1e7 --dt 1e-3 --steps 200 > $O/bench_1025.json 2>$O/bench_1025.err; head -c 400 $O/bench_1025.json; tail -3 $O/bench_1025.err
