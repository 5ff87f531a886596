#!/usr/bin/env python
"""Per-launch SQ counters of the step, attributed through the schedule like tools/pmc_traffic.py:
    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU ... --kernel-trace --output-format csv -d D -o step -- \
        python tools/pmc_step.py --out D/schedule.json
    python tools/pmc_counters.py D D/schedule.json
prints, per tag, every counter divided by SQ_WAVES (per-wave averages)."""
import csv, glob, json, os, sys
from collections import defaultdict

d, sched = sys.argv[1], json.load(open(sys.argv[2]))
steps = sched["steps"]
owner = [i for i, l in enumerate(sched["schedule"]) for _ in range(int(l.get("dispatches", 1)))]   # dispatch -> launch
L = len(owner)
path = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
disp = defaultdict(lambda: defaultdict(float))
for r in csv.DictReader(open(path)):
    if "fillBuffer" in r["Kernel_Name"] or "__amd_rocclr" in r["Kernel_Name"]:
        continue   # blit kernels of the runtime (the memset nodes in front of the single-pass column scans)
    disp[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
ids = sorted(disp)[-steps * L:]
assert len(ids) == steps * L
tags = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
for n, i in enumerate(ids):
    tag = sched["schedule"][owner[n % L]]["tag"]
    cnt[tag] += 1
    for k, v in disp[i].items():
        tags[tag][k] += v
names = sorted({k for t in tags.values() for k in t})
print("tag".ljust(34), " ".join(n.replace("SQ_", "")[:14].rjust(14) for n in names))
for tag, c in tags.items():
    w = c.get("SQ_WAVES", 0) or 1.0
    print(tag[:34].ljust(34), " ".join(f"{(c[n] / cnt[tag] if n == 'SQ_WAVES' else c[n] / w):14.1f}" for n in names))
