#!/bin/bash
# round 3, call 14: column-scan blocks of 48 / 64 rows and the GEMM with the stage barrier in front of its last MFMA group
# (A/B builds: python -m rustpde_mpi_amd.build br48 -DRPDE_COLBR=48 ; ... br64 -DRPDE_COLBR=64 -DRPDE_GEMM_EARLYBAR)
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03n
rm -rf $O; mkdir -p $O
L=rustpde_mpi_amd/librustpde_hip.so
cp $L /tmp/base.so
for v in base br48 br64; do
  [ $v = base ] && cp /tmp/base.so $L || cp rustpde_mpi_amd/librustpde_hip_$v.so $L
  timeout 120 python tools/profile_step.py > $O/profile_$v.txt 2>&1
  echo "== $v"; grep -E "^C4|^C7|^C10|^G1|^G2|^total" $O/profile_$v.txt | cut -c1-100
done
# numerics of the variant that is still installed (br64 + early barrier): GEMM against numpy, solvers, one step
(timeout 250 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "mfma_gemm or test_confined_step or config2 or test_solvers" 2>&1 | tail -4) | tee $O/pytest_br64_eb.txt
cp /tmp/base.so $L
