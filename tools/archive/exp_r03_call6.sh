#!/bin/bash
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03f
rm -rf $O; mkdir -p $O
timeout 300 python tools/trace_phases.py "S3 x: rhs + hholtz-x velx" "S3 x: rhs + hholtz-x vely" "S3 x: rhs + hholtz-x temp" "S2 y: vely" > $O/trace.txt 2>&1
cat $O/trace.txt
timeout 150 python tools/profile_step.py > $O/profile.txt 2>&1
cut -c1-100 $O/profile.txt
