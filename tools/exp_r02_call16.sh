#!/bin/bash
# cache hints (OP_TOUCH) and first-round stagger: per-launch tables with each switch, parity with both on
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02n
rm -rf $O; mkdir -p $O
for cfg in "0 0" "1 0" "0 2" "1 2"; do
  set -- $cfg
  RPDE_TOUCH=$1 RPDE_STAGGER=$2 timeout 200 python tools/profile_step.py > $O/profile_t$1_s$2.txt 2>&1
  echo "== touch $1 stagger $2"; grep -E "^S1|^S2|^S3|^S5|total" $O/profile_t$1_s$2.txt
done
RPDE_TOUCH=1 RPDE_STAGGER=2 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "confined_257 or config2 or confined_step or space_ops" 2>&1 | tail -4
