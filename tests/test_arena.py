"""Device memory bookkeeping of the library (csrc/platform.h ArenaT, C ABI rpde_device_memory / rpde_device_trim /
rpde_arena_check / rpde_arena_selftest): slabs keyed by device, trim on destroy, guard granules.

CPU: the keying logic on a host backend with two pretended devices, run inside the PRODUCT library (a host-only entry
point: no GPU needed).  GPU: an engine under RPDE_ARENA_GUARD=1 (every buffer followed by a guard granule: a WRITE behind
any buffer of a step is counted), the memory going back to the driver when the engine is destroyed, and the same run with
RPDE_ARENA=0 (one hipMalloc per buffer, i.e. every buffer its own mapping: a READ far behind a buffer faults there)."""
import os
import subprocess
import sys

import pytest

import rustpde_mpi_amd as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_arena_keying_selftest_in_the_product_library():
    if not os.path.exists(R.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = R.lib()
    lib.call("rpde_arena_selftest")       # raises with the failed requirement's text


def test_arena_selftest_in_the_emulation_build(emu_lib):
    emu_lib.call("rpde_arena_selftest")


def _child(code, env=None, timeout=600):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)
    sys.stdout.write(r.stdout[-3000:])
    assert r.returncode == 0 and "CHILD-OK" in r.stdout, f"rc {r.returncode}\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}"
    return r.stdout


_STEP = """
import ctypes as C, numpy as np
import rustpde_mpi_amd as R
from oracle import navier as N
lib = R.lib(); assert lib.is_device_build
def mem():
    a, b = C.c_size_t(), C.c_size_t()
    lib.call("rpde_device_memory", 0, C.byref(a), C.byref(b))
    return a.value, b.value
for (nx, ny, periodic) in ((129, 129, False), (257, 193, False), (128, 129, True)):
    mk = (R.Navier2D.new_periodic if periodic else R.Navier2D.new_confined)
    nav = mk(nx, ny, 1e5, 1.0, 0.01, 1.0, "rbc", library=lib, init_random=None)
    ora = (N.Navier2D.new_periodic if periodic else N.Navier2D.new_confined)(nx, ny, 1e5, 1.0, 0.01, 1.0, "rbc", eig_mode="parity")
    for z in (nav, ora):
        z.set_velocity(0.2, 1.0, 1.0); z.set_temperature(0.2, 1.0, 1.0)
    nav.update(5)
    for _ in range(5): ora.update()
    got, want = nav.physical_fields(), ora.physical_fields()
    for k in want:
        e = np.linalg.norm(got[k] - want[k]) / np.linalg.norm(want[k])
        assert e < 1e-10, (nx, ny, k, e)
    v = C.c_long(-1)
    lib.call("rpde_arena_check", C.byref(v))
    print("grid", nx, ny, "periodic" if periodic else "confined", "guard violations", v.value, "slab/used bytes", mem())
    assert v.value == 0, v.value
    if EXPECT_ARENA:
        assert mem()[1] > 0
    del nav, ora
    import gc; gc.collect()
    v = C.c_long(-1)
    lib.call("rpde_arena_check", C.byref(v))       # blocks checked when they were freed
    assert v.value == 0, v.value
    assert mem() == (0, 0), mem()                  # destroy trims: nothing of the engine stays allocated
print("CHILD-OK")
"""


@pytest.mark.gpu
def test_steps_under_guard_granules_and_trim_on_destroy(hip_lib):
    """Three small engines (confined, ragged confined, periodic) step under RPDE_ARENA_GUARD=1: parity with the oracle,
    no guard granule overwritten while they run or when their buffers are freed, and all slabs back at the driver after
    destroy."""
    out = _child("EXPECT_ARENA = True\n" + _STEP, {"RPDE_ARENA_GUARD": "1"})
    assert out.count("guard violations 0") == 3


@pytest.mark.gpu
def test_steps_with_one_mapping_per_buffer(hip_lib):
    """The same with RPDE_ARENA=0: every buffer is its own hipMalloc, so an access far outside a buffer faults instead of
    landing in a neighbour inside a slab (the over-read of the round-4 single-pass scan tables was of that kind)."""
    _child("EXPECT_ARENA = False\n" + _STEP, {"RPDE_ARENA": "0"})
