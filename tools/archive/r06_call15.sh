#!/bin/bash
# Round 6, GPU call 15: chunk-major tables through buffer descriptors (S3, S6, S8): per-launch A/B of two builds (base = the library
# of commit a1616c4, kept as librustpde_hip_base.so), tests of the new build
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06r
rm -rf $O; mkdir -p $O
export RPDE_EIG_CACHE=/tmp/rpde_eig; mkdir -p $RPDE_EIG_CACHE
export RPDE_TOOLS_SPECTRUM=/tmp/spec4097.npy
L=rustpde_mpi_amd/librustpde_hip.so
cp $L /tmp/new.so; cp rustpde_mpi_amd/librustpde_hip_base.so /tmp/base.so
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "confined_step or config2 or whole_line or solvers or round5_ab_switches or s6_derived" 2>&1 | tail -3 | tee $O/pytest.txt
for rep in 1 2 3; do for w in base new; do
  cp /tmp/$w.so $L
  timeout 200 python tools/profile_step.py 2>&1 | grep -E "^S3|^S6|^S8|^total" | sed "s/^/$w rep=$rep  /" >> $O/ab_chunktab.txt
done; done
cat $O/ab_chunktab.txt
for w in base new base new; do cp /tmp/$w.so $L; timeout 120 python tools/ab_step.py | sed "s/^/$w /"; done | tee $O/ab_step.txt
for w in base new; do cp /tmp/$w.so $L
  timeout 300 python bench.py --periodic --nx 4096 --ny 1025 --ra 1e8 --dt 5e-4 --steps 200 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-120 | sed "s/^/config3 $w /" | tee -a $O/bench_cfg.txt
  timeout 300 python bench.py --nx 1025 --ny 1025 --ra 1e7 --dt 1e-3 --steps 200 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-120 | sed "s/^/config2 $w /" | tee -a $O/bench_cfg.txt
done
cp /tmp/new.so $L
