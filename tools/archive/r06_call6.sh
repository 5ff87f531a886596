#!/bin/bash
# Round 6, GPU call 6: A/B of the register budget of S5 / S8 (three instead of four lines per CU: no spills), then the whole GPU suite.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06g
rm -rf $O; mkdir -p $O
export RPDE_EIG_CACHE=/tmp/rpde_eig; mkdir -p $RPDE_EIG_CACHE
export RPDE_TOOLS_SPECTRUM=/tmp/spec4097.npy
for rep in 1 2; do
  for lib in "" librustpde_hip_wpc3.so; do
    RPDE_TOOLS_LIB=$lib timeout 200 python tools/profile_step.py 2>&1 | grep -E "^S5|^S8|^total" | sed "s/^/lib=${lib:-default} rep=$rep  /" >> $O/ab_wpc.txt
  done
done
cat $O/ab_wpc.txt
timeout 2400 python -m pytest tests -m gpu -q -x --durations=12 2>&1 | grep -v "socket.cpp\|amdgpu.ids\|Gloo\] Rank\|mean.h5" | tail -30 > $O/pytest_gpu_full.txt
cat $O/pytest_gpu_full.txt
