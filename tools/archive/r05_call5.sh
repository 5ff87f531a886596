#!/bin/bash
# round 5, GPU call 5: GEMM with the global loads of stage t + 2 issued one MFMA group earlier (A/B, same box)
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r05e; rm -rf $O; mkdir -p $O
export RPDE_TOOLS_SPECTRUM=/tmp/lam4097.npy
for v in base early base2 early2; do
  case $v in base|base2) L="";; early|early2) L="librustpde_hip_gemmearly.so";; esac
  RPDE_TOOLS_LIB=$L timeout 300 python tools/profile_step.py > $O/profile_$v.txt 2>&1
  echo "== $v"; grep -E "G1|G2|total" $O/profile_$v.txt | cut -c1-100
done
