#!/usr/bin/env python
"""Phase times of the single-pass column scan (col_hholtz1_kernel, marks 0 .. 9 of thread 0 of every workgroup):
tools/trace_col1.py [nx ny] [tag ...]   (default tags: hholtz-y, correction-y)
marks: 0 start | 1 ticket | 2 rows loaded + zero-inflow solve (wave 0) | 3 all waves done | 4 aggregate published |
5 arrived, knows whether it is the tile's last | 6 inflow states in LDS (last: sweep + publish; others: wait + 6 rows) |
7 chains with the exact inflow (wave 0) | 8 barrier | 9 rows corrected and stored"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustpde_mpi_amd as R

args = sys.argv[1:]
nx = int(args.pop(0)) if args and args[0].isdigit() else 4097
ny = int(args.pop(0)) if args and args[0].isdigit() else 4097
tags = args or ["hholtz-y", "correction-y"]
nav = R.Navier2D.new_confined(nx, ny, 1e8, 1.0, 2e-4, 1.0, "rbc")
nav.set_velocity(0.2, 1.0, 1.0); nav.set_temperature(0.2, 1.0, 1.0)
nav.update(2)
for tag in tags:
    t, n, span, rows = nav.trace_launch(tag)
    print(f"== {t}: {n} workgroups, kernel span {span:.3f} ms; whole workgroup: mean {rows[0][2]:.0f} p10 {rows[0][3]:.0f} median {rows[0][4]:.0f} p90 {rows[0][5]:.0f} clk")
    for r in rows[2:]:
        print(f"   -> mark {r[0]:2d}: mean {r[2]:8.0f}  p10 {r[3]:8.0f}  median {r[4]:8.0f}  p90 {r[5]:8.0f}")
