#!/bin/bash
# A/B in one call: vec = commit 6d82497, split = + compile-time split phase of the DCT
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02u
rm -rf $O; mkdir -p $O
for round in 1 2; do
for v in vec split; do
  cp tools/ab/$v.so rustpde_mpi_amd/librustpde_hip.so
  timeout 100 python tools/profile_step.py > $O/profile_${v}_$round.txt 2>&1
  echo "== $v $round: $(grep total $O/profile_${v}_$round.txt)"
done
done
for v in vec split; do echo "== $v"; grep -E "^S1|^S2|^S3" $O/profile_${v}_2.txt | cut -c1-72; done
cp tools/ab/split.so rustpde_mpi_amd/librustpde_hip.so
timeout 150 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "space_ops or confined_step or confined_257 or periodic_step or config2 or prandtl" 2>&1 | tail -3
