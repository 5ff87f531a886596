#!/bin/bash
# round 4, call 6: where does a workgroup of the single-pass column scan spend its time (tools/trace_col1.py)
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04f; rm -rf $O; mkdir -p $O
RPDE_COL1_W=16 python tools/trace_col1.py 4097 4097 > $O/trace_4097_w16.txt 2>&1; cat $O/trace_4097_w16.txt
RPDE_COL1_W=8 python tools/trace_col1.py 4097 4097 > $O/trace_4097_w8.txt 2>&1; cat $O/trace_4097_w8.txt
RPDE_COL1_W=8 python tools/trace_col1.py 1025 1025 > $O/trace_1025_w8.txt 2>&1; cat $O/trace_1025_w8.txt
