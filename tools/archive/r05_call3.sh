#!/bin/bash
# round 5, GPU call 3: sharded engines with the overlap order (ranks share the GPU, gloo), same-inputs golden at 1025^2,
# adjoint snapshots are CPU-tested; the allocation-trace replay of the first-step fault (fresh processes, SDMA on / off)
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r05c; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_sharded.py tests/test_gpu_parity.py -m gpu -x -q -s \
  -k "sharded or overlap or config4 or shared_basis" > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
tail -25 $O/pytest.txt
bash tools/fault_repro/run.sh 60 > $O/fault_repro.txt 2>&1; cat $O/fault_repro.txt
