"""Snapshots / restart / callback: the reference's HDF5 layout written and parsed by
csrc/h5lite, checked by an independent parser (tests/h5classic.py) and by a byte-level fixture."""
import hashlib
import os

import numpy as np
import pytest

import rustpde_mpi_amd as R
from tests import checks as K
from tests.h5classic import File


def test_h5lite_roundtrip_and_independent_parser(emu_lib, tmp_path):
    fn = str(tmp_path / "a.h5")
    rng = np.random.default_rng(0)
    want = {"time": np.array([0.25]), "ux/v": rng.standard_normal((5, 7)), "ux/x": np.arange(5.0),
            "temp/vhat_re": rng.standard_normal((3, 4)), "temp/vhat_im": rng.standard_normal((3, 4))}
    for k, v in want.items():
        R.h5.write(fn, k, v, library=emu_lib)          # create, then append dataset by dataset
    assert sorted(R.h5.paths(fn, library=emu_lib)) == sorted(want)
    f = File(fn)
    for k, v in want.items():
        assert np.array_equal(R.h5.read(fn, k, library=emu_lib), v)
        assert np.array_equal(f.datasets[k], v)
    R.h5.write(fn, "time", np.array([0.5]), library=emu_lib)   # overwrite keeps the others
    assert R.h5.read(fn, "time", library=emu_lib)[0] == 0.5 and np.array_equal(File(fn).datasets["ux/v"], want["ux/v"])
    with pytest.raises(R.RpdeError, match="no dataset"):
        R.h5.read(fn, "nope", library=emu_lib)
    with pytest.raises(R.RpdeError, match="not an HDF5 file"):
        open(str(tmp_path / "junk.h5"), "wb").write(b"x" * 200)
        R.h5.read(str(tmp_path / "junk.h5"), "time", library=emu_lib)


def test_h5lite_byte_fixture(emu_lib, tmp_path):
    """The exact bytes of a small file: superblock v0, one root dataset, one group with one dataset."""
    fn = str(tmp_path / "b.h5")
    R.h5.write(fn, "time", np.array([1.5]), library=emu_lib)
    R.h5.write(fn, "g/v", np.arange(6.0).reshape(2, 3), library=emu_lib)
    b = open(fn, "rb").read()
    assert b[:8] == b"\x89HDF\r\n\x1a\n" and b[8:16] == bytes([0, 0, 0, 0, 0, 8, 8, 0])
    assert b[16:20] == bytes([4, 0, 16, 0])                      # group leaf K = 4, internal K = 16
    assert b[96:96 + 48] == np.arange(6.0).tobytes()             # "g/v" first (paths sorted), raw, contiguous, at 96
    assert b[144:152] == np.array([1.5]).tobytes()
    assert int.from_bytes(b[40:48], "little") == len(b)          # end-of-file address
    assert b.count(b"TREE") == 2 and b.count(b"SNOD") == 2 and b.count(b"HEAP") == 2
    assert hashlib.sha256(b).hexdigest() == open(os.path.join(K.GOLDEN, "h5lite_fixture.sha256")).read().strip()


@pytest.mark.parametrize("periodic", [False, True])
def test_snapshot_layout_and_restart(emu_lib, tmp_path, periodic):
    """Navier2D::write / read (navier_io.rs:21-62): layout of SURVEY App. C, bit-exact restart."""
    nav, ora = K.make_pair(emu_lib, periodic, 16 if periodic else 17, 17, 1e4, 1.0, 0.01, 1.0)
    nav.update(3)
    fn = str(tmp_path / "flow.h5")
    nav.write(fn)
    f = File(fn).datasets
    groups = ("ux", "uy", "temp", "pres", "tempbc")
    spec = ("vhat_re", "vhat_im") if periodic else ("vhat",)
    assert sorted(f) == sorted([f"{g}/{d}" for g in groups for d in ("x", "dx", "y", "dy", "v") + spec] +
                               ["time", "ra", "pr", "nu", "ka"])
    x, y = nav.velx.x
    for g, name in zip(groups, ("velx", "vely", "temp", "pres", None)):
        assert np.array_equal(f[g + "/x"], x) and np.array_equal(f[g + "/dx"], x)      # field/io.rs:96-99: dx := x
        assert np.array_equal(f[g + "/y"], y) and np.array_equal(f[g + "/dy"], y)
        if name:
            fld = getattr(nav, name)
            assert np.array_equal(f[g + "/v"], fld.v)
            vh = fld.vhat
            if periodic:
                assert np.array_equal(f[g + "/vhat_re"], vh.real) and np.array_equal(f[g + "/vhat_im"], vh.imag)
            else:
                assert np.array_equal(f[g + "/vhat"], vh)
    # the lift: T = +0.5 at the bottom, -0.5 at the top, linear (boundary_conditions.rs:18-36)
    assert np.allclose(f["tempbc/v"][:, 0], 0.5) and np.allclose(f["tempbc/v"][:, -1], -0.5)
    assert f["time"][0] == nav.get_time() and f["ra"][0] == 1e4 and abs(f["nu"][0] - nav.params["nu"]) == 0
    # restart: a fresh engine continues bit-identically
    ctor = R.Navier2D.new_periodic if periodic else R.Navier2D.new_confined
    nav2 = ctor(nav.nx, nav.ny, 1e4, 1.0, 0.01, 1.0, "rbc", library=emu_lib)
    nav2.read(fn)
    assert nav2.get_time() == nav.get_time()
    nav.update(2); nav2.update(2)
    for k in ("velx", "vely", "temp", "pres"):
        assert np.array_equal(getattr(nav, k).vhat, getattr(nav2, k).vhat), k


@pytest.mark.parametrize("periodic", [False, True])
def test_restart_on_another_resolution(emu_lib, tmp_path, periodic):
    """field/io.rs:151-176: spectral coefficients are truncated / zero-padded; unnormalised Fourier
    coefficients are rescaled by (new_m - 1) / (old_m - 1)."""
    ctor = R.Navier2D.new_periodic if periodic else R.Navier2D.new_confined
    n0 = 16 if periodic else 17
    a = ctor(n0, 17, 1e4, 1.0, 0.01, 1.0, "rbc", library=emu_lib)
    a.set_velocity(0.2, 1.0, 1.0); a.set_temperature(0.2, 1.0, 1.0)
    a.update(2)
    fn = str(tmp_path / "coarse.h5")
    a.write(fn)
    n1 = 32 if periodic else 33
    b = ctor(n1, 33, 1e4, 1.0, 0.01, 1.0, "rbc", library=emu_lib)
    b.read(fn)
    old, new = a.temp.vhat, b.temp.vhat
    norm = (new.shape[0] - 1) / (old.shape[0] - 1) if periodic else 1.0
    assert np.array_equal(new[:old.shape[0], :old.shape[1]], old * norm)
    assert not new[old.shape[0]:].any() and not new[:, old.shape[1]:].any()
    # the interpolated field is the same function: compare on the coarse grid's points that both grids share
    if not periodic:
        assert np.allclose(b.temp.v[::2, ::2], a.temp.v, atol=1e-12)


def test_callback_cadence_and_info_file(emu_lib, tmp_path, monkeypatch, capfd):
    """integrate(..., save_intervall) -> callback(): data/flow{time:0>8.2}.h5 on the write interval and one
    "time nu nuv re" line per callback in data/info.txt (navier_io.rs:84-149, lib.rs:196-203)."""
    monkeypatch.chdir(tmp_path)
    nav, ora = K.make_pair(emu_lib, False, 17, 17, 1e4, 1.0, 0.01, 1.0)
    nav.write_intervall = 0.04
    steps = R.integrate(nav, 0.1, save_intervall=0.02)
    assert steps == 10
    lines = open("data/info.txt").read().split("\n")[:-1]
    assert len(lines) == 5                                   # t = 0.02, 0.04, ..., 0.10
    t, nu, nuv, re = (float(v) for v in lines[-1].split())
    for _ in range(10):
        ora.update()
    assert abs(t - 0.1) < 1e-12 and abs(nu - ora.eval_nu()) < 1e-9 and abs(nuv - ora.eval_nuvol()) < 1e-9
    assert abs(re - ora.eval_re()) < 1e-9 * max(1.0, ora.eval_re())
    assert sorted(os.listdir("data")) == ["flow00000.04.h5", "flow00000.08.h5", "info.txt"]
    assert abs(File("data/flow00000.08.h5").datasets["time"][0] - 0.08) < 1e-12
    out = capfd.readouterr().out
    assert out.count("|div| =") == 5 and "Nu = " in out and "Re = " in out


@pytest.mark.gpu
def test_snapshot_restart_and_callback_on_the_gpu(hip_lib, tmp_path, monkeypatch):
    """The same I/O path through the HIP build: write / independent parse / bit-exact restart, callback
    with the device-side reduction of Nu, Nuvol, Re."""
    monkeypatch.chdir(tmp_path)
    nav, ora = K.make_pair(hip_lib, False, 129, 65, 1e5, 1.0, 0.01, 1.0)
    nav.update(5)
    nav.write("snap.h5")
    f = File("snap.h5").datasets
    assert np.array_equal(f["temp/v"], nav.temp.v) and np.array_equal(f["ux/vhat"], nav.velx.vhat)
    nav2 = R.Navier2D.new_confined(129, 65, 1e5, 1.0, 0.01, 1.0, "rbc", library=hip_lib)
    nav2.read("snap.h5")
    nav.update(3); nav2.update(3)
    for k in ("velx", "vely", "temp", "pres"):
        assert np.array_equal(getattr(nav, k).vhat, getattr(nav2, k).vhat), k
    nav.callback()
    for _ in range(8):
        ora.update()
    t, nu, nuv, re = (float(v) for v in open("data/info.txt").read().split())
    assert abs(nu - ora.eval_nu()) < 1e-9 and abs(nuv - ora.eval_nuvol()) < 1e-9 and abs(re - ora.eval_re()) < 1e-9 * ora.eval_re()
    assert os.path.exists("data/flow00000.08.h5")
