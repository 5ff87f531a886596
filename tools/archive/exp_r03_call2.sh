#!/bin/bash
# round 3, call 2: whole-line backward transform -- one line per workgroup vs persistent workgroups that fetch the next
# line while they transform the current one (RPDE_DCT_PF = 3 / 4 workgroups per CU)
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03b
rm -rf $O; mkdir -p $O
python - > $O/mb.txt 2>&1 <<'PY'
import rustpde_mpi_amd as R
for rep in range(2):
    for w in ("dct_line", "dct_line_pf3", "dct_line_pf4"):
        ms = R.microbench(w, 4097, 4097)
        print(f"{w:14s} {ms:.4f} ms  {8*(4095+4097)*4097/ms/1e6:.0f} GB/s", flush=True)
PY
cat $O/mb.txt
for pf in 3 4; do RPDE_DCT_PF=$pf timeout 120 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "dct_line_backward_4097 or whole_line_kernel" 2>&1 | tail -2; done
