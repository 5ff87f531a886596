#!/bin/bash
# trimmed evidence: rocprofv3 kernel trace of the bench command, then the bench line with the CPU baseline
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r01s; rm -rf $O; mkdir -p $O
cd /tmp
timeout 70 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
cd $R
python tools/trace_by_tag.py $O/trace profiles/r01_schedule.json $O/trace_by_tag.csv 2> $O/trace_by_tag.log
timeout 80 python bench.py --cpu-steps 1 > $O/bench.json 2> $O/bench.err
rm -f $O/*/*.db $O/*/*/*.db
cat $O/trace_by_tag.log; cut -c1-160 $O/bench.json
