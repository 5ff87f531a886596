#!/bin/bash
# first-round stagger by CU arrival order, with the arrivals reported by the traced kernel
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02s
rm -rf $O; mkdir -p $O
for s in 0 2; do
  RPDE_STAGGER=$s timeout 200 python tools/trace_ops.py 4097 4097 "S2 y: velx" "conv_velx" "S1 x" > $O/trace_s$s.txt 2>&1
  echo "== stagger $s"; cat $O/trace_s$s.txt
done
for s in 0 1 2 4; do
  RPDE_STAGGER=$s timeout 200 python tools/profile_step.py > $O/profile_s$s.txt 2>&1
  echo "== stagger $s"; grep -E "^S1|^S2|^S3|total" $O/profile_s$s.txt | cut -c1-72
done
