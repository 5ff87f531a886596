#!/bin/bash
# round 4, call 11: S3 (rhs_line) with the pair in front from the neighbour lane (DPP) instead of a second load: parity, phase trace, times
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04k; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "confined_257 or confined_step or (whole_line_stage and S3) or step_through_the_whole or headline_config_4097" 2>&1 | tail -4
python tools/trace_phases.py 4097 4097 "S3 x: rhs + hholtz-x velx" "S3 x: rhs + hholtz-x vely" > $O/trace_s3.txt 2>&1; cat $O/trace_s3.txt
python tools/profile_step.py 4097 4097 > $O/p4097.txt 2>&1; cat $O/p4097.txt
python tools/profile_step.py 1025 1025 > $O/p1025.txt 2>&1; grep "S3\|total" $O/p1025.txt
python tools/profile_step.py 2049 2049 > $O/p2049.txt 2>&1; grep "S3\|total" $O/p2049.txt
