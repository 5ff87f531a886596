#!/usr/bin/env python
"""Attribute a rocprofv3 --kernel-trace (CSV) of a bench.py run to the launches of the step.

All line programs share three kernel symbols, so the per-symbol --stats table cannot tell `S4` from
`S6`.  The step issues the same L launches in the same order every step (tools/pmc_step.py writes
that schedule); this script finds the longest stretch of the trace that is periodic with period L and
whose kernel kinds match the schedule, and averages the durations per schedule position / tag.

    python tools/trace_by_tag.py <dir with *kernel_trace.csv> <schedule.json> [out.csv]
"""
import csv, glob, json, os, sys
from collections import defaultdict


def kind_of_tag(tag):
    if tag.startswith("T"): return "transpose"
    if tag.startswith("G"): return "gemm"
    if tag.startswith("pseu"): return "set_element"
    if tag.startswith("H"): return "copy2d"
    if tag.startswith("C4 y: hholtz") or tag.startswith("C7"): return "col_hholtz"
    if tag.startswith("C"): return "col_diff"
    return "line_kernel"


def kind_of_kernel(name):
    for k in ("transpose", "gemm", "set_element", "line_kernel", "copy2d", "col_hholtz", "col_diff"):
        if k in name: return k
    if "_line" in name or "hdct_pair" in name or "rfft_pair" in name or "four_rhs" in name: return "line_kernel"   # whole-line kernels: hdct_pair_kernel, conv_line_kernel, rhs_line_kernel ...
    return "other"


def main():
    d, sched = sys.argv[1], json.load(open(sys.argv[2]))
    # one entry per kernel dispatch: a column scan is 5 (Helmholtz) or 3 (derivative) kernels
    tags = [l["tag"] for l in sched["schedule"] for _ in range(int(l.get("dispatches", 1)))]
    launches = {}
    for l in sched["schedule"]:
        launches[l["tag"]] = launches.get(l["tag"], 0) + 1
    want = [kind_of_tag(t) for t in tags]
    L = len(tags)
    path = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    # (the memset nodes in front of the single-pass column scans are blit kernels of the runtime: not launches of the step)
    rows = sorted((r for r in csv.DictReader(open(path)) if "fillBuffer" not in r["Kernel_Name"] and "__amd_rocclr" not in r["Kernel_Name"]),
                  key=lambda r: int(r["Start_Timestamp"]))
    kinds = [kind_of_kernel(r["Kernel_Name"]) for r in rows]
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3 for r in rows]   # us
    best = (0, 0)
    i = 0
    while i + L <= len(rows):
        if kinds[i:i + L] == want:
            n = 0
            while i + (n + 1) * L <= len(rows) and kinds[i + n * L:i + (n + 1) * L] == want:
                n += 1
            if n > best[1]: best = (i, n)
            i += n * L
        else:
            i += 1
    start, nsteps = best
    assert nsteps > 0, "no stretch of the trace matches the schedule"
    acc = defaultdict(list)
    for s in range(nsteps):
        for p in range(L):
            acc[tags[p]].append(dur[start + s * L + p])
    step_us = sum(sum(v) for v in acc.values()) / nsteps
    out = [("tag", "launches_per_step", "launches", "mean_us", "min_us", "max_us", "us_per_step", "share")]
    for t in dict.fromkeys(tags):
        v = acc[t]
        per = tags.count(t) // launches[t]          # kernels per launch of this tag
        # mean / min / max per LAUNCH (the kernels of a column scan added up)
        lv = [sum(v[i:i + per]) for i in range(0, len(v), per)] if per > 1 else v
        if per > 1:   # acc appends dispatch by dispatch within a step: regroup per step first
            lv = []
            nl = launches[t]
            for s_ in range(nsteps):
                chunk = v[s_ * nl * per:(s_ + 1) * nl * per]
                lv += [sum(chunk[i:i + per]) for i in range(0, len(chunk), per)]
        out.append((t, launches[t], len(lv), f"{sum(lv)/len(lv):.2f}", f"{min(lv):.2f}", f"{max(lv):.2f}",
                    f"{sum(v)/nsteps:.1f}", f"{sum(v)/nsteps/step_us:.4f}"))
    out.append(("TOTAL kernel time per step", L, nsteps * L, "", "", "", f"{step_us:.1f}", "1.0"))
    w = csv.writer(open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout)
    w.writerows(out)
    print(f"# {nsteps} steps of {L} launches matched; kernel time {step_us/1e3:.3f} ms/step", file=sys.stderr)


if __name__ == "__main__":
    main()
