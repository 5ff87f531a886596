#!/bin/bash
# round 3, call 7 (new session): where the step stands at HEAD 063eeef -- per-launch table, per-barrier trace of every
# whole-line kernel, the parity tests of the whole-line kernels
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03g
rm -rf $O; mkdir -p $O
timeout 200 python tools/profile_step.py > $O/profile.txt 2>&1
cut -c1-110 $O/profile.txt
timeout 300 python tools/trace_phases.py "S1 x" "S2 y: velx" "S2 y: conv_velx" "S2 y: conv_temp" "S3 x: rhs + hholtz-x velx" "S3 x: rhs + hholtz-x vely" > $O/trace.txt 2>&1
cat $O/trace.txt
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "dct_line or whole_line or conv_line or headline or rhs_line" 2>&1 | tail -4 | tee $O/pytest.txt
