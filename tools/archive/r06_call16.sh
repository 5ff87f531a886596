#!/bin/bash
# Round 6, GPU call 16: rows and tables of S8 / S6 through buffer descriptors too; S8 at four lines per CU (variant build): per-launch A/B of
# three builds (base = commit a1616c4, new, new + RPDE_S8_WPC=4)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06s
rm -rf $O; mkdir -p $O
export RPDE_EIG_CACHE=/tmp/rpde_eig; mkdir -p $RPDE_EIG_CACHE
export RPDE_TOOLS_SPECTRUM=/tmp/spec4097.npy
L=rustpde_mpi_amd/librustpde_hip.so
cp $L /tmp/new.so; cp rustpde_mpi_amd/librustpde_hip_base.so /tmp/base.so; cp rustpde_mpi_amd/librustpde_hip_s8w4.so /tmp/s8w4.so
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "confined_step or config2 or whole_line or solvers or round5_ab_switches or s6_derived" 2>&1 | tail -3 | tee $O/pytest.txt
for rep in 1 2 3; do for w in base new s8w4; do
  cp /tmp/$w.so $L
  timeout 200 python tools/profile_step.py 2>&1 | grep -E "^S3|^S6|^S8|^total" | sed "s/^/$w rep=$rep  /" >> $O/ab_chunktab.txt
done; done
cat $O/ab_chunktab.txt
for w in base new s8w4 base new s8w4; do cp /tmp/$w.so $L; timeout 120 python tools/ab_step.py | sed "s/^/$w /"; done | tee $O/ab_step.txt
cp /tmp/new.so $L
