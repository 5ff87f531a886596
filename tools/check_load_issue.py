#!/usr/bin/env python
"""Static check of the line kernels' ISA: inside one barrier phase, no global load may sit behind an
`s_waitcnt vmcnt(0)` that follows an earlier load of the same phase (the compiler branches around a predicated
load and waits inside the branch -- ten dependent HBM round trips per thread instead of one).
usage: tools/kernel_resources.sh > /dev/null; tools/check_load_issue.py [/tmp/rpde_kernels.s]"""
import re, subprocess, sys
path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/rpde_kernels.s"
txt = open(path).read().split("\n")
kern, name = {}, None
for l in txt:
    m = re.match(r"^(_ZN4rpde11line_kernel\S+):", l)
    if m:
        name = m.group(1); kern[name] = []
    elif name:
        kern[name].append(l.strip())
        if l.strip().startswith("s_endpgm"): name = None
bad_total = 0
for name, body in kern.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    seg, nseg, bad, worst = [], 0, 0, 0
    for t in body + ["s_barrier"]:
        if not t or t[0] in ";.": continue
        seg.append(t)
        if t.startswith("s_barrier"):
            gl = [j for j, u in enumerate(seg) if u.startswith("global_load") or u.startswith("scratch_load")]
            ser = sum(1 for a, b in zip(gl, gl[1:]) if any(u.startswith("s_waitcnt vmcnt(0)") for u in seg[a:b]))
            if ser >= 3: bad += 1; worst = max(worst, ser)
            nseg += 1; seg = []
    bad_total += bad
    print(f"{bad:3d} phases with >= 3 serialised loads (worst {worst:2d}) of {nseg:4d}  {dem[22:100]}")
sys.exit(1 if bad_total else 0)
