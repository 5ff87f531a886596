#!/usr/bin/env python
"""Attribute rocprofv3 per-dispatch FETCH_SIZE / WRITE_SIZE to the launches of the step.

The last steps*L kernel dispatches of a tools/pmc_step.py run are the steps (L = launches per step,
one kernel each on a single GPU); the kernel-name sequence must repeat with period L, otherwise
the script stops.  Corrections (MI355X_MICROARCH.md "HBM", and calibrated here on the `copy`
microbench whose traffic is known exactly: 4096 lines x 4097 f64 read and written once):
  * both counters are in KiB;
  * FETCH_SIZE counts 128-byte read requests as 64 bytes on gfx950 -> x2 (measured 65 875 KiB for
    131 104 KiB read with 8 B/lane coalesced loads);
  * WRITE_SIZE is exact (131 072 KiB for 131 104 KiB written).
"""
import argparse
import csv
import glob
import hashlib
import json
import os


def csrc_sha256():
    """sha256 over the kernel / engine sources the counters were collected on (bench.py marks a traffic file stale when
    the sources it runs on hash differently)."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rustpde_mpi_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(root)):
        if f.endswith((".h", ".cc")):
            h.update(f.encode())
            h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()


def per_position(dirname, counter, steps, L):
    path = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    assert path, f"no counter_collection.csv under {dirname}"
    disp = {}
    for r in csv.DictReader(open(path[0])):
        if r["Counter_Name"] != counter or "fillBuffer" in r["Kernel_Name"] or "__amd_rocclr" in r["Kernel_Name"]:
            continue   # (blit kernels of the runtime: the memset nodes in front of the single-pass column scans)
        d = int(r["Dispatch_Id"])
        disp[d] = (r["Kernel_Name"], disp.get(d, ("", 0.0))[1] + float(r["Counter_Value"]))
    seq = [disp[k] for k in sorted(disp)][-steps * L:]
    assert len(seq) == steps * L, f"{len(seq)} dispatches, expected {steps}*{L}"
    for s in range(1, steps):
        for i in range(L):
            assert seq[i][0] == seq[s * L + i][0], "dispatch sequence is not periodic in the step length"
    return [(seq[i][0], sum(seq[s * L + i][1] for s in range(steps)) / steps * 1024.0) for i in range(L)]


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--fetch", required=True)
    p.add_argument("--write", required=True)
    p.add_argument("--schedule", required=True)
    p.add_argument("--out", required=True)
    a = p.parse_args()
    sch = json.load(open(a.schedule))
    steps = sch["steps"]
    # a launch of the schedule is one kernel dispatch, except the column scans (5 / 3 kernels)
    nd = [int(l.get("dispatches", 1)) for l in sch["schedule"]]
    L = sum(nd)
    fe_d = per_position(a.fetch, "FETCH_SIZE", steps, L)
    wr_d = per_position(a.write, "WRITE_SIZE", steps, L)
    fe, wr, p = [], [], 0
    for n in nd:   # fold the dispatches of a launch into one entry
        fe.append((fe_d[p][0], sum(x[1] for x in fe_d[p:p + n])))
        wr.append((wr_d[p][0], sum(x[1] for x in wr_d[p:p + n])))
        p += n
    tags = {}
    for i, l in enumerate(sch["schedule"]):
        t = tags.setdefault(l["tag"], {"launches_per_step": 0, "algorithmic_bytes": l["bytes"], "flops": l["flops"],
                                       "kernel": fe[i][0].split("(")[0][:96], "read_bytes": 0.0, "write_bytes": 0.0})
        t["launches_per_step"] += 1
        t["read_bytes"] += 2.0 * fe[i][1]
        t["write_bytes"] += wr[i][1]
    for tag, t in tags.items():
        n = t["launches_per_step"]
        t["read_bytes"] /= n
        t["write_bytes"] /= n
        t["traffic_bytes"] = t["read_bytes"] + t["write_bytes"]
        t["traffic_over_algorithmic"] = t["traffic_bytes"] / t["algorithmic_bytes"] if t["algorithmic_bytes"] else None
        t["event_ms_per_launch"] = sch["event_ms"].get(tag)
    out = {"workload": sch["workload"], "steps_averaged": steps, "csrc_sha256": csrc_sha256(),
           "commit": os.environ.get("EVIDENCE_COMMIT", "").strip() or None,
           "corrections": "KiB -> bytes; FETCH_SIZE x2 (gfx950 128-B requests tallied as 64 B); WRITE_SIZE x1",
           "per_launch": tags,
           "step_total_traffic_bytes": sum(t["traffic_bytes"] * t["launches_per_step"] for t in tags.values()),
           "step_total_algorithmic_bytes": sum(t["algorithmic_bytes"] * t["launches_per_step"] for t in tags.values())}
    json.dump(out, open(a.out, "w"), indent=1)
    for tag, t in tags.items():
        print(f"{tag:34s} x{t['launches_per_step']}  alg {t['algorithmic_bytes']/1e6:8.1f} MB  "
              f"traffic {t['traffic_bytes']/1e6:8.1f} MB  ratio {t['traffic_over_algorithmic'] or 0:.2f}")


if __name__ == "__main__":
    main()
