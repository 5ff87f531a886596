#!/bin/bash
# Round-5 evidence, headline part, on the FINAL sources (the first evidence run, tools/evidence_r05.sh, predates the LNSE
# solver and the completed 4097^2 same-inputs golden: its per-configuration bench lines and kernel stats of configs 2 / 3 / 5
# stay valid -- the kernels of the time step did not change after it): the full GPU test suite (with the
# same-inputs goldens at 4097^2 and 2049^2 and the LNSE tests), PMC traffic and SQ counter passes, the bench line, rocprofv3 kernel trace + stats of the bench command.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05f
rm -rf $O; mkdir -p $O
cat $R/.evidence_commit > $O/commit.txt 2>/dev/null
export EVIDENCE_COMMIT=$(cat $O/commit.txt 2>/dev/null)
(timeout 1800 python -m pytest tests -m gpu -q -s --durations=8 2>&1 | grep -v "socket.cpp\|amdgpu.ids\|Gloo\] Rank" | tail -150) > $O/pytest_gpu_final.txt
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o step -- python $R/tools/pmc_step.py --out $O/schedule.json > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o step -- python $R/tools/pmc_step.py --out $O/schedule.json > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/sq -o step -- python $R/tools/pmc_step.py --out $O/schedule.json > $O/pmc_sq.log 2>&1
cd $R
python tools/pmc_traffic.py --fetch $O/fetch --write $O/write --schedule $O/schedule.json --out $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
python tools/pmc_counters.py $O/sq $O/schedule.json > $O/sq_counters.txt 2>&1
cp $O/pmc_traffic.json profiles/r05_pmc_traffic.json     # bench.py picks roofline.traffic up from here
python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
cd $R
python tools/trace_by_tag.py $O/trace $O/schedule.json $O/trace_by_tag.csv 2> $O/trace_by_tag.log
python tools/profile_step.py > $O/profile_step.txt 2>&1
python bench.py --no-cpu-baseline --bc hc --steps 30 > $O/bench_hc.json 2> $O/bench_hc.err      # the blocked column solve
RPDE_HC_BLOCKED=0 python bench.py --no-cpu-baseline --bc hc --steps 30 > $O/bench_hc_serial.json 2> $O/bench_hc_serial.err
bash tools/kernel_resources.sh > $O/kernel_resources.txt 2>/dev/null
rm -f $O/*/*.db $O/*/*/*.db
find $O -name '*kernel_trace.csv' -size +8M -delete
rm -f $O/fetch/*counter_collection.csv $O/write/*counter_collection.csv $O/sq/*counter_collection.csv $O/*/*/*counter_collection.csv
tail -30 $O/pytest_gpu_final.txt | cut -c1-200; tail -c 2500 $O/bench.json; cat $O/trace_by_tag.log; head -30 $O/pmc_traffic.txt
