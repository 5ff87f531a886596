#!/bin/bash
# Round-4 evidence on one MI355X, all from ONE commit: GPU tests, PMC traffic passes, SQ counters, the bench line,
# rocprofv3 kernel trace + stats of the same bench command, kernel resources, other configs.  Writes gpurun_out/r04/ ;
# the summaries are then copied to profiles/r04_* (tools/collect_r04.sh).
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04
rm -rf $O; mkdir -p $O
cat $R/.evidence_commit > $O/commit.txt 2>/dev/null   # written by the caller (git rev-parse HEAD): the GPU box has no .git
export EVIDENCE_COMMIT=$(cat $O/commit.txt 2>/dev/null)
(timeout ${PYTEST_TIMEOUT:-1500} python -m pytest tests -m gpu -q -s --durations=12 ${PYTEST_K:+-k "$PYTEST_K"} 2>&1 | tail -30) > $O/pytest_gpu.txt
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o step -- python $R/tools/pmc_step.py --out $O/schedule.json > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o step -- python $R/tools/pmc_step.py --out $O/schedule.json > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/sq -o step -- python $R/tools/pmc_step.py --out $O/schedule.json > $O/pmc_sq.log 2>&1
cd $R
python tools/pmc_traffic.py --fetch $O/fetch --write $O/write --schedule $O/schedule.json --out $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
python tools/pmc_counters.py $O/sq $O/schedule.json > $O/sq_counters.txt 2>&1
cp $O/pmc_traffic.json profiles/r04_pmc_traffic.json     # bench.py picks roofline.traffic up from here
python bench.py ${BENCH_FLAGS-} > $O/bench.json 2> $O/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
cd $R
python tools/trace_by_tag.py $O/trace $O/schedule.json $O/trace_by_tag.csv 2> $O/trace_by_tag.log
python tools/profile_step.py > $O/profile_step.txt 2>&1
python bench.py --no-cpu-baseline --nx 1025 --ny 1025 --ra 1e7 --dt 1e-3 --steps 200 > $O/bench_1025.json 2>/dev/null
python bench.py --no-cpu-baseline --periodic --nx 4096 --ny 1025 --steps 100 > $O/bench_periodic.json 2>/dev/null
python bench.py --no-cpu-baseline --periodic --nx 16384 --ny 2049 --aspect 8 --ra 1e9 --dt 1e-4 --steps 30 > $O/bench_config5_1gpu.json 2>/dev/null
bash tools/kernel_resources.sh > $O/kernel_resources.txt 2>/dev/null
# the other single-GPU configurations: rocprofv3 kernel stats of the bench command and the per-launch table (HIP events)
for cfg in "config2:--nx 1025 --ny 1025 --ra 1e7 --dt 1e-3 --steps 200:1025 1025" "config3:--periodic --nx 4096 --ny 1025 --steps 100:4096 1025 periodic" "config5:--periodic --nx 16384 --ny 2049 --aspect 8 --ra 1e9 --dt 1e-4 --steps 30:16384 2049 periodic"; do
  name=${cfg%%:*}; rest=${cfg#*:}; flags=${rest%%:*}; prof=${rest#*:}
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$name -o bench -- python $R/bench.py --no-cpu-baseline $flags > /dev/null 2> $O/trace_$name.err)
  cp $O/trace_$name/*/*kernel_stats.csv $O/kernel_stats_$name.csv 2>/dev/null || cp $O/trace_$name/*kernel_stats.csv $O/kernel_stats_$name.csv 2>/dev/null
  rm -rf $O/trace_$name
  python tools/profile_step.py $prof > $O/profile_step_$name.txt 2>&1
done
python tools/trace_col1.py 4097 4097 > $O/trace_col1.txt 2>&1
rm -f $O/*/*.db $O/*/*/*.db
find $O -name '*kernel_trace.csv' -size +8M -delete
rm -f $O/fetch/*counter_collection.csv $O/write/*counter_collection.csv $O/sq/*counter_collection.csv $O/*/*/*counter_collection.csv  # large; summaries kept
cat $O/pytest_gpu.txt | tail -20; tail -c 3000 $O/bench.json; cat $O/trace_by_tag.log; head -14 $O/trace/*/*kernel_stats.csv $O/trace/*kernel_stats.csv 2>/dev/null; head -45 $O/pmc_traffic.txt
