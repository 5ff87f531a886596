#!/bin/bash
# Round 2, GPU call 2: parity diagnostics at 4097, microbench of the wave-serial scans, bench (slim on/off),
# the changed GPU tests.  Writes gpurun_out/r02b/.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r02b
rm -rf $O; mkdir -p $O
python tools/microbench.py 4097 4096 > $O/mb.txt 2>&1
RPDE_SLIM=1 python tools/microbench.py 4097 4096 > $O/mb_slim.txt 2>&1
cat $O/mb.txt $O/mb_slim.txt
python bench.py --no-cpu-baseline > $O/bench_nocpu.json 2> $O/bench_nocpu.err
RPDE_SLIM=1 python bench.py --no-cpu-baseline > $O/bench_nocpu_slim.json 2> $O/bench_nocpu_slim.err
python - <<'PY'
import json
for f in ("bench_nocpu", "bench_nocpu_slim"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r02b/{f}.json") if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d["ms_per_step_update_plus_exit"], d["transform_pass"]["frac_of_hbm_peak"])
        for p in d["phases"]: print("   ", p)
    except Exception as e:
        print(f, "failed", e)
PY
python tools/profile_step.py > $O/profile_step.txt 2>&1; cat $O/profile_step.txt
python tools/diag_parity.py 4097 > $O/diag.txt 2>&1; cat $O/diag.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "bench_sizes or headline or eigenbasis or exit_flag or prandtl or full_eigen or confined_257 or config2" 2>&1 | tail -30 > $O/pytest_sel.txt
cat $O/pytest_sel.txt
