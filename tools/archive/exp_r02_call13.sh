#!/bin/bash
# what can v_mfma_f64_16x16x4_f64 deliver?  workgroups of 4 waves, 1 / 2 / 4 per CU, accumulators in VGPRs / AGPRs
export TMPDIR=/tmp
for w in mfma_peak mfma_peak_a; do for blocks in 256 512 1024; do
python - <<PY
import sys; sys.path.insert(0,'.')
import rustpde_mpi_amd as R
iters=4000
ms=R.microbench("$w", $blocks, iters, 5)
fl=$blocks*4*iters*16*2048
print("$w blocks=$blocks: %.3f ms  %.2f TFLOP/s" % (ms, fl/ms/1e9))
PY
done; done
