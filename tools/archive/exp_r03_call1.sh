#!/bin/bash
# round 3, call 1: first hardware run of conv_line (whole convection term per y-line), then the per-launch table of the
# step with the line programs (RPDE_CONV_LINE=0) and with the whole-line convection kernel (=1)
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03a
rm -rf $O; mkdir -p $O
timeout 200 python tools/check_conv_line_gpu.py > $O/check_conv.txt 2>&1; echo "check rc=$?" >> $O/check_conv.txt
tail -12 $O/check_conv.txt
RPDE_CONV_LINE=0 timeout 150 python tools/profile_step.py > $O/profile_conv0.txt 2>&1
RPDE_CONV_LINE=1 timeout 150 python tools/profile_step.py > $O/profile_conv1.txt 2>&1
echo "--- conv0"; cut -c1-100 $O/profile_conv0.txt
echo "--- conv1"; grep -E "^S2|total" $O/profile_conv1.txt | cut -c1-100
