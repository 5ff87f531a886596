#!/bin/bash
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03e
rm -rf $O; mkdir -p $O
timeout 300 python tools/trace_phases.py "S3 x: rhs + hholtz-x velx" "S3 x: rhs + hholtz-x vely" "S2 y: vely" > $O/trace.txt 2>&1
cat $O/trace.txt
