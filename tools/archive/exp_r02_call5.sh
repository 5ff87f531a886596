#!/bin/bash
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r02e
rm -rf $O; mkdir -p $O
for v in 0 1 2 3; do
  echo "== GEMM variant $v" >> $O/mb.txt
  RPDE_GEMM_VARIANT=$v python tools/microbench_one.py gemm_nt 2048 4095 50 >> $O/mb.txt 2>&1
  RPDE_GEMM_VARIANT=$v python tools/microbench_one.py gemm_nn 2048 4095 50 >> $O/mb.txt 2>&1
done
cat $O/mb.txt
RPDE_GEMM_VARIANT=2 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "mfma_gemm" 2>&1 | tail -2
RPDE_S1_MERGE=1 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "confined_step or confined_257" 2>&1 | tail -2
RPDE_S1_MERGE=1 python tools/profile_step.py > $O/profile_step_merge.txt 2>&1; head -8 $O/profile_step_merge.txt; tail -2 $O/profile_step_merge.txt
