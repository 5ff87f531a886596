#!/bin/bash
# first-round stagger by CU arrival order (HW_ID ticket): per-launch tables
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02o
rm -rf $O; mkdir -p $O
for s in 0 1 2 3 102; do
  RPDE_STAGGER=$s timeout 200 python tools/profile_step.py > $O/profile_s$s.txt 2>&1
  echo "== stagger $s"; grep -E "^S|total" $O/profile_s$s.txt | cut -c1-75
done
RPDE_STAGGER=2 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "confined_257 or config2 or confined_step" 2>&1 | tail -3
