#!/bin/bash
# Round 6, GPU call 11: G2 accumulated transposed (128-byte stores instead of 32-byte pieces): bit-identity, A/B timing per kernel and per step
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06n
rm -rf $O; mkdir -p $O
export RPDE_EIG_CACHE=/tmp/rpde_eig; mkdir -p $RPDE_EIG_CACHE
export RPDE_TOOLS_SPECTRUM=/tmp/spec4097.npy
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "transposed_accumulation or eight_waves or mfma_gemm" 2>&1 | tail -3 | tee $O/pytest.txt
for rep in 1 2 3; do for w in 0 1; do
  RPDE_GEMM_CTSWAP=$w timeout 200 python tools/profile_step.py 2>&1 | grep -E "^G1|^G2|^total" | sed "s/^/ctswap=$w rep=$rep  /" >> $O/ab_gemm_ctswap.txt
done; done
cat $O/ab_gemm_ctswap.txt
(RPDE_GEMM_CTSWAP=0 timeout 120 python tools/ab_step.py; timeout 120 python tools/ab_step.py; RPDE_GEMM_CTSWAP=0 timeout 120 python tools/ab_step.py; timeout 120 python tools/ab_step.py) | tee $O/ab_step.txt
