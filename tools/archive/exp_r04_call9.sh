#!/bin/bash
# round 4, call 9 (= call 7 again with a consistent build): column scans with lane-distributed row coefficients (v_readlane),
# LDS-resident transfers, the y-derivative in one pass: parity, phase trace, times
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04i; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "one_pass or confined_257 or confined_step or space_ops" 2>&1 | tail -4
RPDE_COL1_W=16 python tools/trace_col1.py 4097 4097 > $O/trace_4097_w16.txt 2>&1; cat $O/trace_4097_w16.txt
RPDE_COL1_W=8 python tools/trace_col1.py 4097 4097 > $O/trace_4097_w8.txt 2>&1; cat $O/trace_4097_w8.txt
for cfg in "w8:X=1" "w16:RPDE_COL1_W=16" "off:RPDE_COL_ONEPASS=0"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs python tools/profile_step.py 4097 4097 > $O/p4097_$name.txt 2>&1
  echo "== 4097 $name"; grep "column scan\|S1 x\|total" $O/p4097_$name.txt
done
for cfg in "w8:X=1" "w16:RPDE_COL1_W=16" "off:RPDE_COL_ONEPASS=0"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs python tools/profile_step.py 1025 1025 > $O/p1025_$name.txt 2>&1
  echo "== 1025 $name"; grep "column scan\|S1 x\|total" $O/p1025_$name.txt
  env $envs python tools/profile_step.py 2049 2049 > $O/p2049_$name.txt 2>&1
  echo "== 2049 $name"; grep "column scan\|S1 x\|total" $O/p2049_$name.txt
done
