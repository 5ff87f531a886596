#!/bin/bash
# Round 6, GPU call 2: (1) what clock and MFMA-pipe occupancy do the two GEMM launches really run at?  (2) the reference's
# criterion benches on the engine.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06b
rm -rf $O; mkdir -p $O
export RPDE_EIG_CACHE=/tmp/rpde_eig; mkdir -p $RPDE_EIG_CACHE
cd /tmp
rocprofv3 -L > $O/counters_avail_full.txt 2>&1
grep -i -E "MFMA|GRBM_GUI|GRBM_COUNT|BUSY_CU|SQ_BUSY|LEVEL_WAVES|SQ_CYCLES|INSTS_VALU_MFMA|INST_LEVEL" $O/counters_avail_full.txt | cut -c1-220 | sort -u | head -80 > $O/counters_avail.txt
rm -f $O/counters_avail_full.txt
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/clk -o step -- python $R/tools/pmc_step.py --out $O/schedule.json > $O/pmc_clk.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/mfma -o step -- python $R/tools/pmc_step.py --out $O/schedule.json > $O/pmc_mfma.log 2>&1
cd $R
python tools/pmc_counters.py $O/clk $O/schedule.json > $O/clk_counters.txt 2>&1
python tools/pmc_counters.py $O/mfma $O/schedule.json > $O/mfma_counters.txt 2>&1
# kernel durations of the same run (kernel trace): GUI_ACTIVE / duration = clock
python - $O <<'PY' > $O/clock_per_kernel.txt 2>&1
import csv, glob, sys, os
from collections import defaultdict
O = sys.argv[1]
tr = glob.glob(os.path.join(O, "clk", "**", "*kernel_trace.csv"), recursive=True)[0]
cc = glob.glob(os.path.join(O, "clk", "**", "*counter_collection.csv"), recursive=True)[0]
dur = {}
for r in csv.DictReader(open(tr)):
    dur[int(r["Dispatch_Id"])] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
cnt = defaultdict(dict)
for r in csv.DictReader(open(cc)):
    cnt[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
agg = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for i, (name, ns) in dur.items():
    c = cnt.get(i, {})
    if "GRBM_GUI_ACTIVE" not in c: continue
    a = agg[name[:60]]
    a[0] += 1; a[1] += ns; a[2] += c["GRBM_GUI_ACTIVE"]; a[3] += c.get("GRBM_COUNT", 0.0)
for name, (n, ns, gui, cnt_) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{name:60s} n {n:4d}  avg {ns / n / 1e3:9.1f} us  GUI_ACTIVE/ns {gui / ns:7.4f} (GHz if it counts shader clocks over all SEs: see the ratio between kernels)  GRBM_COUNT/ns {cnt_ / ns:7.4f}")
PY
cat $O/clock_per_kernel.txt | cut -c1-200
cat $O/clk_counters.txt | cut -c1-200 | grep -E "tag|G1|G2|S6|T1"
cat $O/mfma_counters.txt | cut -c1-200 | grep -E "tag|G1|G2"
tail -3 $O/pmc_mfma.log
timeout 900 python tools/bench_criterion.py --out $O/criterion.json > $O/criterion.txt 2>&1
tail -30 $O/criterion.txt | cut -c1-250
rm -f $O/*/*.db $O/*/*/*.db
find $O -name '*kernel_trace.csv' -size +8M -delete
find $O -name '*counter_collection.csv' -size +8M -delete
