#!/usr/bin/env python
"""Where the time of a line program goes: per-op shader-clock durations inside the kernel
(Program::trace, rpde_navier2d_trace_launch).  tools/trace_ops.py [nx ny] [tag ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustpde_mpi_amd as R

args = sys.argv[1:]
nx = int(args.pop(0)) if args and args[0].isdigit() else 4097
ny = int(args.pop(0)) if args and args[0].isdigit() else 4097
tags = args or ["S1 x", "S2 y: vel", "conv_velx", "conv_temp", "S3 x: rhs + hholtz-x velx", "S5", "S6", "S7", "S8", "S9"]
nav = R.Navier2D.new_confined(nx, ny, 1e8, 1.0, 2e-4, 1.0, "rbc")
nav.set_velocity(0.2, 1.0, 1.0); nav.set_temperature(0.2, 1.0, 1.0)
nav.update(); nav.update()
for tag in tags:
    try:
        t, n, span, rows = nav.trace_launch(tag)
    except Exception as e:   # tag not in this configuration
        print(f"-- {tag}: {e}")
        continue
    tot = rows[-1][4]
    print(f"== {t}: {n} workgroups, kernel span {span:.3f} ms, program median {tot:.0f} clk")
    print(f"   {'ip':>2s} {'op':8s} {'mean':>8s} {'p10':>8s} {'median':>8s} {'p90':>8s} {'share':>6s}")
    for ip, op, mean, p10, med, p90 in rows:
        print(f"   {ip:2d} {op:8s} {mean:8.0f} {p10:8.0f} {med:8.0f} {p90:8.0f} {med / max(tot, 1):6.3f}")
