#!/bin/bash
# per-op clocks inside the line programs, pair DCT on / off
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02h
rm -rf $O; mkdir -p $O
RPDE_DCT_PAIR=0 timeout 300 python tools/trace_ops.py > $O/trace_single.txt 2>&1; cat $O/trace_single.txt
RPDE_DCT_PAIR=1 timeout 300 python tools/trace_ops.py "S1 x" "S2 y: vel" conv_velx > $O/trace_pair.txt 2>&1; cat $O/trace_pair.txt
