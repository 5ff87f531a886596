#!/bin/bash
# gpurun_out/r05f -> profiles/r05_* (the headline files of the final sources; the first evidence run's files of the other
# configurations stay, its headline files move to profiles/r05_first_run/)
O=gpurun_out/r05f; P=profiles
mkdir -p $P/r05_first_run
for f in r05_bench.json r05_bench_profiled.json r05_kernel_stats.csv r05_trace_by_tag.csv r05_pmc_traffic.txt r05_sq_counters.txt r05_profile_step.txt r05_schedule.json r05_kernel_resources.txt r05_commit.txt; do
  [ -f $P/$f ] && git mv -f $P/$f $P/r05_first_run/$f 2>/dev/null || mv -f $P/$f $P/r05_first_run/$f 2>/dev/null
done
c=$(cut -c1-7 $O/commit.txt 2>/dev/null)
cp $O/commit.txt $P/r05_commit.txt
cp $O/pytest_gpu_final.txt $P/r05_pytest_gpu_full_final_${c}.txt
[ -f $P/r05_bench_hc.json ] && git mv -f $P/r05_bench_hc.json $P/r05_first_run/r05_bench_hc.json
for f in bench bench_profiled bench_hc bench_hc_serial pmc_traffic schedule; do cp $O/$f.json $P/r05_$f.json 2>/dev/null; done
for f in pmc_traffic sq_counters profile_step kernel_resources; do cp $O/$f.txt $P/r05_$f.txt 2>/dev/null; done
cp $O/trace_by_tag.csv $P/r05_trace_by_tag.csv 2>/dev/null
cp $O/trace/*/*kernel_stats.csv $P/r05_kernel_stats.csv 2>/dev/null || cp $O/trace/*kernel_stats.csv $P/r05_kernel_stats.csv 2>/dev/null
ls $P | grep r05
