#!/bin/bash
# whole-line DCT kernel (csrc/dct_line.h: 256 threads per line, four workgroups per CU) for the pure transforms of S2:
# RPDE_DCT_LINE=0 (line program) against =1 in one call, then parity on the GPU
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02v
rm -rf $O; mkdir -p $O
for v in 0 1; do
  RPDE_DCT_LINE=$v timeout 60 python tools/profile_step.py > $O/profile_dl$v.txt 2>&1
  echo "== RPDE_DCT_LINE=$v"; grep -E "^S2 y: vel|total" $O/profile_dl$v.txt | cut -c1-80
done
timeout 60 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "dct_line_backward_4097 or whole_line_kernel" 2>&1 | tail -3
