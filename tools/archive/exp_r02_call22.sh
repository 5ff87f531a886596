#!/bin/bash
# A/B in one call: base = HEAD (tables.so), vec = 16-byte line loads / stores
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02t
rm -rf $O; mkdir -p $O
for round in 1 2; do
for v in tables vec; do
  cp tools/ab/$v.so rustpde_mpi_amd/librustpde_hip.so
  timeout 200 python tools/profile_step.py > $O/profile_${v}_$round.txt 2>&1
  echo "== $v $round: $(grep total $O/profile_${v}_$round.txt)"
done
done
for v in tables vec; do echo "== $v"; grep -E "^S|^C" $O/profile_${v}_2.txt | cut -c1-72; done
cp tools/ab/vec.so rustpde_mpi_amd/librustpde_hip.so
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "not headline and not config5" 2>&1 | tail -3
