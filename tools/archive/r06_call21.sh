#!/bin/bash
# Round 6, GPU call 21: Navier2DNonLin::update on the fused schedule: GPU tests of the 8f-4 solvers, bench lines fused / generic
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06x
rm -rf $O; mkdir -p $O
export RPDE_EIG_CACHE=/tmp/rpde_eig; mkdir -p $RPDE_EIG_CACHE
timeout 1500 python -m pytest tests/test_adjoint.py -m gpu -q -x > $O/pytest_adjoint.txt 2>&1; grep -E "passed|failed|error" $O/pytest_adjoint.txt | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "confined_step or config2 or conv or periodic_config3 or forked" > $O/pytest_parity.txt 2>&1; grep -E "passed|failed|error" $O/pytest_parity.txt | tail -3
for fz in 1 0; do
  RPDE_LNSE_FUSED=$fz timeout 300 python bench.py --solver nonlin --nx 1025 --ny 1025 --ra 1e5 --dt 0.01 --steps 20 --warmup 3 --cpu-steps 2 2>> $O/nonlin.err | grep '^{' > $O/bench_nonlin_1025_fused$fz.json
  RPDE_LNSE_FUSED=$fz timeout 300 python bench.py --solver nonlin --nx 4097 --ny 4097 --ra 1e8 --dt 2e-4 --steps 5 --warmup 1 --no-cpu-baseline 2>> $O/nonlin.err | grep '^{' > $O/bench_nonlin_4097_fused$fz.json
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06x/bench_nonlin_*.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], round(d["value"], 2), round(d["ms_per_step"], 4), (d.get("parity") or {}).get("ok"), (d.get("parity") or {}).get("rel_l2"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 $O/nonlin.err
