#!/bin/bash
# per-phase clocks inside the line programs
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02i
rm -rf $O; mkdir -p $O
timeout 300 python tools/trace_ops.py > $O/trace_phases.txt 2>&1; cat $O/trace_phases.txt
timeout 200 python tools/profile_step.py > $O/profile_step.txt 2>&1; tail -1 $O/profile_step.txt
