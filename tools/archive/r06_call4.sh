#!/bin/bash
# Round 6, GPU call 4: the solvers of SURVEY 8f-4 after the generic operators lost their per-call allocations and host waits:
# GPU parity tests, first bench lines (with CPU oracle + parity legs), a rocprofv3 kernel summary of one adjoint update.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${RPDE_CALL_DIR:-r06d}
rm -rf $O; mkdir -p $O
export RPDE_EIG_CACHE=/tmp/rpde_eig; mkdir -p $RPDE_EIG_CACHE
timeout 900 python -m pytest tests/test_adjoint.py tests/test_general_lengths.py -m gpu -q -x -k "adjoint or lnse or nonlin or gradient" 2>&1 | grep -v "mean.h5\|Gloo" | tail -6 > $O/pytest_adjoint.txt; cat $O/pytest_adjoint.txt
for sv in adjoint lnse lnse_adjoint nonlin; do
  timeout 600 python bench.py --solver $sv --nx 1025 --ny 1025 --ra 1e5 --dt 0.01 --steps 20 --warmup 3 --cpu-steps 2 2> $O/bench_${sv}_1025.err | grep '^{' > $O/bench_${sv}_1025.json
  python -c "import json,sys; d=json.load(open('$O/bench_${sv}_1025.json')); print('$sv', round(d['ms_per_step'],3), 'ms/update; cpu', round(1e3/d['cpu_baseline']['value'],1), 'ms; parity', d['parity']['rel_l2'], d['parity']['ok'])" 2>&1 | tail -1
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_adjoint -o adj -- python $R/bench.py --solver adjoint --nx 1025 --ny 1025 --ra 1e5 --dt 0.01 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_adjoint_profiled.json 2> $O/bench_adjoint_profiled.err
cd $R
python - $O <<'PY'
import csv, glob, sys, os
O = sys.argv[1]
f = glob.glob(os.path.join(O, "trace_adjoint", "**", "*kernel_stats.csv"), recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    with open(os.path.join(O, "adjoint_1025_kernel_stats.csv"), "w") as out:
        out.write(open(f[0]).read())
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    calls = sum(int(r["Calls"]) for r in rows)
    print("kernels", len(rows), "launches", calls, "GPU ms total", tot / 1e6, "(6 updates)")
    for r in rows[:12]:
        print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"]) / 1e3:9.1f} us  {r["Percentage"]:>6s} %')
PY
timeout 600 python bench.py --solver adjoint --nx 4097 --ny 4097 --ra 1e8 --dt 1e-4 --steps 3 --warmup 1 --no-cpu-baseline 2> $O/bench_adjoint_4097.err | grep '^{' > $O/bench_adjoint_4097.json; cut -c1-400 $O/bench_adjoint_4097.json; tail -2 $O/bench_adjoint_4097.err
timeout 600 python bench.py --solver lnse --nx 4097 --ny 4097 --ra 1e8 --dt 2e-4 --steps 5 --warmup 1 --no-cpu-baseline 2> $O/bench_lnse_4097.err | grep '^{' > $O/bench_lnse_4097.json; cut -c1-400 $O/bench_lnse_4097.json; tail -2 $O/bench_lnse_4097.err
rm -f $O/*/*.db $O/*/*/*.db; find $O -name '*kernel_trace.csv' -size +4M -delete
