#!/bin/bash
# round 3, call 4: S3 as one kernel per field (rhs_line.h): A/B parity on the GPU, then the step table
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03d
rm -rf $O; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "whole_line_stage_equals and (S3 or S1) or dct_line_backward_4097 or conv_line_4097" 2>&1 | tail -5
RPDE_S3_LINE=1 timeout 150 python tools/profile_step.py > $O/profile_s3_1.txt 2>&1
RPDE_S3_LINE=0 timeout 150 python tools/profile_step.py > $O/profile_s3_0.txt 2>&1
echo "--- S3 whole-line"; grep -E "^S3|^S9|total" $O/profile_s3_1.txt | cut -c1-100
echo "--- S3 line program"; grep -E "^S3|^S9|total" $O/profile_s3_0.txt | cut -c1-100
