#!/usr/bin/env python
"""Shader clocks between consecutive barriers of a traced launch (whole-line kernels and line programs alike):
tools/trace_phases.py [nx ny] tag [tag ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustpde_mpi_amd as R

args = sys.argv[1:]
nx = int(args.pop(0)) if args and args[0].isdigit() else 4097
ny = int(args.pop(0)) if args and args[0].isdigit() else 4097
nav = R.Navier2D.new_confined(nx, ny, 1e8, 1.0, 2e-4, 1.0, "rbc")
nav.set_velocity(0.2, 1.0, 1.0); nav.set_temperature(0.2, 1.0, 1.0)
nav.update(2)
for tag in args:
    t, n, span, rows = nav.trace_launch(tag)
    print(f"== {t}: {n} workgroups, kernel span {span:.3f} ms, per-workgroup median {rows[0][4]:.0f} clk (p10 {rows[0][3]:.0f}, p90 {rows[0][5]:.0f})")
    print("   interval medians:", " ".join(f"{r[4]:.0f}" for r in rows[2:]))
