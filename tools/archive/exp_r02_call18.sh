#!/bin/bash
# one workgroup per CU (a third, unused LDS slot) against two: is a DCT line program bound by the CU or by latency?
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02p
rm -rf $O; mkdir -p $O
for s in 2 3; do
  RPDE_S2_SLOTS=$s timeout 200 python tools/profile_step.py > $O/profile_slots$s.txt 2>&1
  echo "== slots $s"; grep -E "^S2 y: vel|total" $O/profile_slots$s.txt | cut -c1-80
  RPDE_S2_SLOTS=$s timeout 200 python tools/trace_ops.py 4097 4097 "S2 y: velx" "S2 y: vely" > $O/trace_slots$s.txt 2>&1
  cat $O/trace_slots$s.txt
done
