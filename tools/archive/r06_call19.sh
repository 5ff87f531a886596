#!/bin/bash
# Round 6, GPU call 19: full GPU suite on the new default build (guarded factor loads on the half-length core, Navier2DLnse fused);
# A/B of four builds: goff (unguarded), new, h4 (half-length convection term at a 128-register budget, ~30 spilled),
# half (the 4097-point term on the half-length core, 155 registers) -- tests of `half` first
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06v
rm -rf $O; mkdir -p $O
export RPDE_EIG_CACHE=/tmp/rpde_eig; mkdir -p $RPDE_EIG_CACHE
export RPDE_TOOLS_SPECTRUM=/tmp/spec4097.npy
L=rustpde_mpi_amd/librustpde_hip.so
cp $L /tmp/new.so; for v in goff h4 half; do cp rustpde_mpi_amd/librustpde_hip_$v.so /tmp/$v.so; done
timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 > $O/pytest_full.txt 2>&1; grep -E "passed|failed|error" $O/pytest_full.txt | tail -3
cp /tmp/half.so $L
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "conv or confined_step or whole_line or space_ops" > $O/pytest_half.txt 2>&1; grep -E "passed|failed|error" $O/pytest_half.txt | tail -3
for rep in 1 2 3; do
  for w in new half; do cp /tmp/$w.so $L
    timeout 200 python tools/profile_step.py 2>&1 | grep -E "^S2|^total" | sed "s/^/4097 $w rep=$rep  /" >> $O/ab_conv.txt
  done
  for w in goff new h4; do cp /tmp/$w.so $L
    timeout 100 python tools/profile_step.py 1025 1025 2>&1 | grep -E "^S2|^total" | sed "s/^/1025 $w rep=$rep  /" >> $O/ab_conv.txt
    timeout 100 python tools/profile_step.py 4096 1025 periodic 2>&1 | grep -E "^S2|^total" | sed "s/^/4096x1025 $w rep=$rep  /" >> $O/ab_conv.txt
    timeout 100 python tools/profile_step.py 16384 2049 periodic 2>&1 | grep -E "^S2|^total" | sed "s/^/16384x2049 $w rep=$rep  /" >> $O/ab_conv.txt
  done
done
cat $O/ab_conv.txt
for w in new half new half; do cp /tmp/$w.so $L; timeout 120 python tools/ab_step.py | sed "s/^/$w /"; done | tee $O/ab_step.txt
for w in new h4 new h4; do cp /tmp/$w.so $L
  timeout 300 python bench.py --periodic --nx 4096 --ny 1025 --ra 1e8 --dt 5e-4 --steps 200 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-120 | sed "s/^/config3 $w /" | tee -a $O/bench_cfg.txt
  timeout 300 python bench.py --nx 1025 --ny 1025 --ra 1e7 --dt 1e-3 --steps 200 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-120 | sed "s/^/config2 $w /" | tee -a $O/bench_cfg.txt
done
cp /tmp/new.so $L
