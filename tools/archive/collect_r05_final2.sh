#!/bin/bash
# gpurun_out/r05g -> profiles/r05_* (the headline files of the final sources; the first evidence run's headline files move to
# profiles/r05_first_run/, its files of the other configurations stay)
O=gpurun_out/r05g; P=profiles
mkdir -p $P/r05_first_run
for f in NONE; do   # (the first run was moved once; later calls overwrite the headline files in place)
  [ -f $O/../.moved_$f ] && continue
  [ -f $P/$f ] && (git mv -f $P/$f $P/r05_first_run/$f 2>/dev/null || mv -f $P/$f $P/r05_first_run/$f)
done
c=$(cut -c1-7 $O/commit.txt 2>/dev/null)
cp $O/commit.txt $P/r05_commit.txt
for f in bench bench_profiled pmc_traffic schedule bench_1025 bench_hc; do [ -s $O/$f.json ] && cp $O/$f.json $P/r05_$f.json; done
for f in pmc_traffic sq_counters lds_counters profile_step kernel_resources ab_step ab_step_1025 ab_s6_keep laps; do [ -s $O/$f.txt ] && cp $O/$f.txt $P/r05_$f.txt; done
[ -s $O/pytest_gpu_new.txt ] && cp $O/pytest_gpu_new.txt $P/r05_pytest_gpu_new_${c}.txt
[ -s $O/pytest_gpu_more.txt ] && cp $O/pytest_gpu_more.txt $P/r05_pytest_gpu_more_${c}.txt
[ -s $O/trace_by_tag.csv ] && cp $O/trace_by_tag.csv $P/r05_trace_by_tag.csv
cp $O/trace/*/*kernel_stats.csv $P/r05_kernel_stats.csv 2>/dev/null || cp $O/trace/*kernel_stats.csv $P/r05_kernel_stats.csv 2>/dev/null
[ -s $O/defaults.txt ] && cp $O/defaults.txt $P/r05_defaults_from_ab.txt
ls $P | grep r05
