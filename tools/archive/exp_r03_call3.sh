#!/bin/bash
# round 3, call 3: half-length transform core (hdct_line.h) against dct_line.h: microbench, operator parity, step table
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03c
rm -rf $O; mkdir -p $O
python - > $O/mb.txt 2>&1 <<'PY'
import os, subprocess, sys
code = """
import rustpde_mpi_amd as R
for rep in range(2):
    ms = R.microbench("dct_line", 4097, 4097)
    print("dct_line RPDE_HDCT=%s  %.4f ms  %.0f GB/s" % (__import__('os').environ.get('RPDE_HDCT'), ms, 8*(4095+4097)*4097/ms/1e6), flush=True)
"""
for h in ("0", "1"):
    env = dict(os.environ, RPDE_HDCT=h)
    print(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout, flush=True)
PY
cat $O/mb.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "dct_line_backward_4097 or whole_line_kernel or conv_line_4097" 2>&1 | tail -3
RPDE_HDCT=1 timeout 150 python tools/profile_step.py > $O/profile_h1.txt 2>&1
RPDE_HDCT=0 timeout 150 python tools/profile_step.py > $O/profile_h0.txt 2>&1
echo "--- hdct=1"; grep -E "^S1|^S2|total" $O/profile_h1.txt | cut -c1-100
echo "--- hdct=0"; grep -E "^S1|^S2|total" $O/profile_h0.txt | cut -c1-100
