#!/bin/bash
# Round 6, GPU call 7: Navier2DAdjoint with its forward step on the fused schedule: parity tests, A/B against the generic composition.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06h
rm -rf $O; mkdir -p $O
export RPDE_EIG_CACHE=/tmp/rpde_eig; mkdir -p $RPDE_EIG_CACHE
timeout 900 python -m pytest tests/test_adjoint.py -m gpu -q -x -k "adjoint" 2>&1 | grep -v "mean.h5\|Gloo" | tail -5 > $O/pytest_adjoint.txt; cat $O/pytest_adjoint.txt
for n in 1025 4097; do
  for f in 1 0; do
    if [ $n = 1025 ]; then A="--ra 1e5 --dt 0.01 --steps 20 --warmup 3"; else A="--ra 1e8 --dt 1e-4 --steps 4 --warmup 2"; fi
    RPDE_ADJOINT_FUSED=$f timeout 600 python bench.py --solver adjoint --nx $n --ny $n $A --no-cpu-baseline 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('adjoint $n fused=$f', round(d['ms_per_step'],3), 'ms/update')" | tee -a $O/ab_adjoint_fused.txt
  done
done
timeout 600 python bench.py --solver adjoint --nx 1025 --ny 1025 --ra 1e5 --dt 0.01 --steps 20 --warmup 3 --cpu-steps 2 2>> $O/err.txt | grep '^{' > $O/bench_adjoint_1025.json
python -c "import json; d=json.load(open('$O/bench_adjoint_1025.json')); print(d['ms_per_step'], d['parity'])"
timeout 600 python bench.py --solver adjoint --nx 4097 --ny 4097 --ra 1e8 --dt 1e-4 --steps 4 --warmup 2 --no-cpu-baseline 2>> $O/err.txt | grep '^{' > $O/bench_adjoint_4097.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_adjoint -o adj -- python $R/bench.py --solver adjoint --nx 1025 --ny 1025 --ra 1e5 --dt 0.01 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_adjoint_profiled.json 2> $O/bench_adjoint_profiled.err
cd $R
f=$(find $O/trace_adjoint -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/adjoint_1025_kernel_stats.csv && head -8 $O/adjoint_1025_kernel_stats.csv | cut -c1-160
rm -f $O/*/*.db $O/*/*/*.db; find $O -name '*kernel_trace.csv' -size +4M -delete
tail -3 $O/err.txt
