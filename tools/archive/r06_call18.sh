#!/bin/bash
# Round 6, GPU call 18: (a) the factor loads of the convection terms behind a run-time condition (conv_line 167 -> 136 registers,
# hconv_line 400 -> 154: three waves per SIMD instead of one at 1025-point lines): per-launch A/B of three builds -- goff (the loads
# unconditional: rounds 4 - 6), new (default), c4 (the 4097-point term at four workgroups per CU, 128 registers, 8 spilled);
# (b) Navier2DLnse::update on the fused schedule: tests, bench lines fused / generic
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06u
rm -rf $O; mkdir -p $O
export RPDE_EIG_CACHE=/tmp/rpde_eig; mkdir -p $RPDE_EIG_CACHE
export RPDE_TOOLS_SPECTRUM=/tmp/spec4097.npy
L=rustpde_mpi_amd/librustpde_hip.so
cp $L /tmp/new.so; cp rustpde_mpi_amd/librustpde_hip_goff.so /tmp/goff.so; cp rustpde_mpi_amd/librustpde_hip_c4.so /tmp/c4.so
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_adjoint.py -m gpu -q -x -k "confined_step or config2 or whole_line or conv or periodic or space_ops or lnse or adjoint_step" 2>&1 | tail -4 | tee $O/pytest.txt
for rep in 1 2 3; do for w in goff new c4; do
  cp /tmp/$w.so $L
  timeout 200 python tools/profile_step.py 2>&1 | grep -E "^S2|^total" | sed "s/^/4097 $w rep=$rep  /" >> $O/ab_conv.txt
  timeout 100 python tools/profile_step.py 1025 1025 2>&1 | grep -E "^S2|^total" | sed "s/^/1025 $w rep=$rep  /" >> $O/ab_conv.txt
  timeout 100 python tools/profile_step.py 4096 1025 periodic 2>&1 | grep -E "^S2|^total" | sed "s/^/4096x1025 $w rep=$rep  /" >> $O/ab_conv.txt
done; done
cat $O/ab_conv.txt
for w in goff new c4 goff new c4; do cp /tmp/$w.so $L; timeout 120 python tools/ab_step.py | sed "s/^/$w /"; done | tee $O/ab_step.txt
for w in goff new goff new; do cp /tmp/$w.so $L
  timeout 300 python bench.py --periodic --nx 4096 --ny 1025 --ra 1e8 --dt 5e-4 --steps 200 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-120 | sed "s/^/config3 $w /" | tee -a $O/bench_cfg.txt
  timeout 300 python bench.py --nx 1025 --ny 1025 --ra 1e7 --dt 1e-3 --steps 200 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-120 | sed "s/^/config2 $w /" | tee -a $O/bench_cfg.txt
done
timeout 300 python bench.py --periodic --nx 16384 --ny 2049 --ra 1e9 --dt 1e-4 --aspect 8 --steps 30 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-120 | sed "s/^/config5 new /" | tee -a $O/bench_cfg.txt
cp /tmp/goff.so $L
timeout 300 python bench.py --periodic --nx 16384 --ny 2049 --ra 1e9 --dt 1e-4 --aspect 8 --steps 30 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-120 | sed "s/^/config5 goff /" | tee -a $O/bench_cfg.txt
cp /tmp/new.so $L
# (b) LNSE fused / generic
for fz in 1 0; do
  RPDE_LNSE_FUSED=$fz timeout 300 python bench.py --solver lnse --nx 1025 --ny 1025 --ra 1e5 --dt 0.01 --steps 20 --warmup 3 --cpu-steps 2 2>> $O/lnse.err | grep '^{' | cut -c1-400 | sed "s/^/lnse1025 fused=$fz /" | tee -a $O/bench_lnse.txt
  RPDE_LNSE_FUSED=$fz timeout 300 python bench.py --solver lnse --nx 4097 --ny 4097 --ra 1e8 --dt 2e-4 --steps 5 --warmup 1 --no-cpu-baseline 2>> $O/lnse.err | grep '^{' | cut -c1-400 | sed "s/^/lnse4097 fused=$fz /" | tee -a $O/bench_lnse.txt
done
tail -5 $O/lnse.err
