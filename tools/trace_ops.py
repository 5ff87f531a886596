#!/usr/bin/env python
"""Where the time of a line program goes: shader clocks per op and per barrier phase inside the kernel
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
    tot = rows[0][4]
    print(f"== {t}: {n} workgroups, kernel span {span:.3f} ms, program median {tot:.0f} clk")
    # group the marks by op: a row with id >= 0 closes the previous op
    ops, cur = [], None
    for id_, name, mean, p10, med, p90 in rows[1:]:
        if cur is not None:
            cur["phases"].append((med, p10, p90))
        if id_ >= 0:
            cur = {"ip": id_, "name": name, "phases": []}
            ops.append(cur)
    # rows[1] is the first op's start mark (delta from nothing): the loop above handles it because cur is None there
    for op in ops[:-1]:
        tot_op = sum(p[0] for p in op["phases"])
        ph = " ".join(f"{p[0]:.0f}" for p in op["phases"])
        print(f"   {op['ip']:2d} {op['name']:8s} {tot_op:8.0f} {tot_op / max(tot, 1):6.3f}   phases: {ph}")
