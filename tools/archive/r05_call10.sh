#!/bin/bash
# round 5, GPU call 10: which LDS layout of the GEMM's operand stages is faster PER KERNEL (G1 / G2), on one box, alternating
# (the whole-step A/B of the evidence call could not tell: 0.03 ms inside a run-to-run spread of 0.06)
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r05h2; rm -rf $O; mkdir -p $O
export RPDE_EIG_CACHE=/tmp/rpde_eig; mkdir -p $RPDE_EIG_CACHE
for rep in 1 2 3; do
  for lay in 1 0; do
    RPDE_GEMM_LDS=$lay timeout 120 python tools/profile_step.py 2>&1 | grep -E "^G1|^G2|^total" | sed "s/^/lay=$lay rep=$rep  /" >> $O/gemm_ab.txt
  done
done
cat $O/gemm_ab.txt
