#!/bin/bash
# round 5, GPU call 2: adjoint engine + tensor Helmholtz parity, the A/B switches (bit-identical), operator tests of the
# touched kernels, then A/B timing of the GEMM loop and the S1 split on the bench workload (per-launch HIP-event profile)
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r05b; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_adjoint.py tests/test_gpu_parity.py tests/test_arena.py -m gpu -x -q -s \
  -k "adjoint or hholtz or descent or round5 or mfma_gemm or headline_config_4097 or conv_line or dct_line_backward or guard or step_through_the_whole_line" > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
tail -30 $O/pytest.txt
export RPDE_TOOLS_SPECTRUM=/tmp/lam4097.npy
for v in base gemm_r4 s1_split base2; do
  case $v in base|base2) E="";; gemm_r4) E="RPDE_GEMM_R4=1";; s1_split) E="RPDE_S1_SPLIT=1";; esac
  env $E timeout 300 python tools/profile_step.py > $O/profile_$v.txt 2>&1
  echo "== $v"; cat $O/profile_$v.txt | cut -c1-110
done
