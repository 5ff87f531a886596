#!/bin/bash
# round 5, GPU call 4: LNSE parity, the independent goldens with full-field norms, remaining sharded test; A/B of two suspects
# for the +5 % of S3 / S5 / S8 in the evidence run: zero-fill DPP in the scans, slab sizes of the arena (array placement)
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r05d; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_adjoint.py tests/test_gpu_parity.py tests/test_sharded.py -m gpu -q -s \
  -k "lnse or independent_reference or extended or config4 or shared_basis_1025" 2>&1 | grep -v "socket.cpp\|amdgpu.ids\|Gloo\] Rank" > $O/pytest.txt
echo "pytest rc $?" >> $O/pytest.txt
tail -30 $O/pytest.txt | cut -c1-250
export RPDE_TOOLS_SPECTRUM=/tmp/lam4097.npy
for v in base dppold slab1g base2 dppold2 slab1g2; do
  case $v in base|base2) L="";; dppold|dppold2) L="librustpde_hip_dppold.so";; slab1g|slab1g2) L="librustpde_hip_slab1g.so";; esac
  RPDE_TOOLS_LIB=$L timeout 300 python tools/profile_step.py > $O/profile_$v.txt 2>&1
  echo "== $v"; grep -E "S1|S3|S5|S8|C4 y: hh|C7|conv_velx|total" $O/profile_$v.txt | cut -c1-100
done
