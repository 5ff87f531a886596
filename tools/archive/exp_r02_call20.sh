#!/bin/bash
# A/B in one call (boxes differ by several per cent): base = HEAD, tables = table reads batched (STEN, MV3, TABDIV, rfft twiddles),
# full = tables + unpredicated line loads
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02r
rm -rf $O; mkdir -p $O
for round in 1 2; do
for v in base tables full; do
  cp tools/ab/$v.so rustpde_mpi_amd/librustpde_hip.so
  timeout 200 python tools/profile_step.py > $O/profile_${v}_$round.txt 2>&1
  echo "== $v $round: $(grep total $O/profile_${v}_$round.txt)"
done
done
for v in base tables full; do echo "== $v"; grep -E "^S|^C" $O/profile_${v}_2.txt | cut -c1-72; done
