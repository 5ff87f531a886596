#!/bin/bash
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r02f
rm -rf $O; mkdir -p $O
python tools/microbench.py 4097 4096 > $O/mb.txt 2>&1; cat $O/mb.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "confined_step or periodic_step or confined_257 or prandtl or test_solvers or space_ops" 2>&1 | tail -3
python tools/profile_step.py > $O/profile_step.txt 2>&1; cat $O/profile_step.txt
