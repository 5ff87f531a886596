#!/bin/bash
# S1 (value + x-derivative of the state lines) through the whole-line kernel: one per-launch table with RPDE_S1_LINE=1
# (compare with 0.238 - 0.250 ms per launch of the line program in every earlier call), then the operator parity on the GPU
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02w
rm -rf $O; mkdir -p $O
RPDE_S1_LINE=1 timeout 40 python tools/profile_step.py > $O/profile_s1line.txt 2>&1
grep -E "^S1|^S2 y: vel|^T1|total" $O/profile_s1line.txt | cut -c1-80
timeout 25 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "dct_line_backward_4097" 2>&1 | tail -2
