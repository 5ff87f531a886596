"""The C ABI from a real C99 translation unit: include/rustpde_hip.h must be valid C, and a host
that only knows the header gets the same numbers as the Python mirror."""
import os
import subprocess

import numpy as np
import pytest

import rustpde_mpi_amd as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_host", "host.c")


def _build_and_run(lib, tmp_path, nx, ny, steps):
    exe = str(tmp_path / "host")
    libdir, libfile = os.path.split(lib.path)
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
           SRC, "-o", exe, "-L", libdir, "-l:" + libfile, "-Wl,-rpath," + libdir, "-lm"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    out = subprocess.run([exe, str(nx), str(ny), str(steps)], check=True, capture_output=True, text=True).stdout.split()
    return {out[i]: float(out[i + 1]) for i in range(0, len(out), 2)}


def _check(lib, tmp_path, nx, ny, steps):
    got = _build_and_run(lib, tmp_path, nx, ny, steps)
    nav = R.Navier2D.new_confined(nx, ny, 1e5, 1.0, 0.01, 1.0, "rbc", library=lib)
    nav.set_velocity(0.2, 1.0, 1.0)
    nav.set_temperature(0.2, 1.0, 1.0)
    nav.update(steps)
    t = nav.temp.v
    assert abs(got["time"] - steps * 0.01) < 1e-14
    assert got["exit"] == 0.0
    assert abs(got["sum"] - t.sum()) < 1e-9 * max(1.0, abs(t.sum()))
    assert abs(got["sumsq"] - (t * t).sum()) < 1e-12 * (t * t).sum()


def test_c_host_emulation_build(emu_lib, tmp_path):
    _check(emu_lib, tmp_path, 17, 33, 4)


@pytest.mark.gpu
def test_c_host_hip_build(hip_lib, tmp_path):
    _check(hip_lib, tmp_path, 129, 65, 10)
