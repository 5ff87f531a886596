#!/bin/bash
# round 5, last GPU call (2.6 GPU-minutes were left): the fast part of tests/test_gpu_parity.py on the final sources -- operator-level
# transforms / stencils / derivatives / solvers, the small confined and periodic steps, transposes, the 1025-point paths
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r05i; rm -rf $O; mkdir -p $O
export RPDE_EIG_CACHE=/tmp/rpde_eig; mkdir -p $RPDE_EIG_CACHE
(timeout 100 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --durations=5 \
   -k "space_ops or solvers or known_answers or transpose or confined_step or confined_257 or periodic_step or 1025 or prandtl or exit" 2>&1 \
   | grep -v "socket.cpp\|amdgpu.ids\|Gloo\] Rank" | tail -20) > $O/pytest_gpu_fast.txt
cat $O/pytest_gpu_fast.txt
