"""libhdf5-side check of the snapshot files, through the REAL HDF5 C library of the image (ctypes binding
tests/libhdf5.py; HDF5 1.10.6 under /opt/conda/lib -- found in round 5).  The consumers of these files are plot/plot2d.py:30-54
(h5py) and src/io/read_write_hdf5.rs:38-188 (the hdf5 crate): both are this library.

csrc/h5lite writes and parses the classic HDF5 structures by hand, and the checker of the other tests (tests/h5classic.py) is a
second reading of the same specification by the same author; here the library itself (i) opens h5lite snapshots and must find
every dataset with the values the engine holds, (ii) writes the same layout with its defaults (symbol-table groups, contiguous
datasets: what the reference's hdf5 crate produces) and an engine restarts from it bit for bit, (iii) appends to an h5lite file
that h5lite then reads back, and (iv) h5dump -- the library's own tool -- accepts the file.  tests/test_h5py_interop.py is the
same through h5py where that is installed; this module skips only on a box without any libhdf5."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import rustpde_mpi_amd as R
from tests import checks as K
from tests import libhdf5 as H

pytestmark = pytest.mark.skipif(H.load() is None, reason="no libhdf5 on this box")

GROUPS = (("ux", "velx"), ("uy", "vely"), ("temp", "temp"), ("pres", "pres"))


def test_the_library_is_the_real_one():
    assert H.version()[0] == 1 and H.version()[1] >= 8, H.version()


@pytest.mark.parametrize("periodic", [False, True])
def test_libhdf5_reads_an_h5lite_snapshot(emu_lib, tmp_path, periodic):
    nav, _ = K.make_pair(emu_lib, periodic, 16 if periodic else 17, 17, 1e4, 1.0, 0.01, 1.0)
    nav.update(3)
    fn = str(tmp_path / "flow.h5")
    nav.write(fn)
    with H.File(fn, "r") as f:                                 # what plot2d.py / read_write_hdf5.rs do
        got = f.datasets()
    assert sorted(got) == sorted(R.h5.paths(fn, library=emu_lib))
    for path, arr in got.items():
        assert np.array_equal(arr, R.h5.read(fn, path, library=emu_lib)), path
    x, y = nav.velx.x
    for g, name in GROUPS:
        assert np.array_equal(got[g + "/v"], getattr(nav, name).v)
        assert np.array_equal(got[g + "/x"], x) and np.array_equal(got[g + "/y"], y)
        vh = getattr(nav, name).vhat
        if periodic:
            assert np.array_equal(got[g + "/vhat_re"], vh.real) and np.array_equal(got[g + "/vhat_im"], vh.imag)
        else:
            assert np.array_equal(got[g + "/vhat"], vh)
    assert got["time"].shape == (1,) and got["time"][0] == nav.get_time()


@pytest.mark.parametrize("periodic", [False, True])
def test_engine_restarts_from_a_libhdf5_written_snapshot(emu_lib, tmp_path, periodic):
    """The same layout written by libhdf5 with its defaults: Navier2D::read must restore the state bit for bit."""
    nav, _ = K.make_pair(emu_lib, periodic, 16 if periodic else 17, 17, 1e4, 1.0, 0.01, 1.0)
    nav.update(3)
    fn = str(tmp_path / "by_libhdf5.h5")
    x, y = nav.velx.x
    with H.File(fn, "w") as f:
        for g, name in GROUPS:
            fld = getattr(nav, name)
            for k, v in (("x", x), ("dx", x), ("y", y), ("dy", y), ("v", fld.v)):
                f.write(g + "/" + k, v)
            vh = fld.vhat
            if periodic:
                f.write(g + "/vhat_re", np.ascontiguousarray(vh.real))
                f.write(g + "/vhat_im", np.ascontiguousarray(vh.imag))
            else:
                f.write(g + "/vhat", vh)
        for k, v in (("time", nav.get_time()), ("ra", 1e4), ("pr", 1.0), ("nu", nav.params["nu"]), ("ka", nav.params["ka"])):
            f.write(k, np.array([v]))
    ctor = R.Navier2D.new_periodic if periodic else R.Navier2D.new_confined
    nav2 = ctor(nav.nx, nav.ny, 1e4, 1.0, 0.01, 1.0, "rbc", library=emu_lib)
    nav2.read(fn)
    assert nav2.get_time() == nav.get_time()
    for _, name in GROUPS:
        assert np.array_equal(getattr(nav2, name).vhat, getattr(nav, name).vhat), name
    nav.update(2); nav2.update(2)
    for _, name in GROUPS:
        assert np.array_equal(getattr(nav2, name).v, getattr(nav, name).v), name


def test_libhdf5_appends_to_an_h5lite_file(emu_lib, tmp_path):
    fn = str(tmp_path / "mixed.h5")
    a = np.arange(12.0).reshape(3, 4)
    R.h5.write(fn, "g/a", a, library=emu_lib)
    R.h5.write(fn, "time", np.array([0.5]), library=emu_lib)
    with H.File(fn, "a") as f:
        f.write("g/b", 2.0 * a)
        f.write("extra", np.array([7.0]))
    assert np.array_equal(R.h5.read(fn, "g/b", library=emu_lib), 2.0 * a)
    assert np.array_equal(R.h5.read(fn, "g/a", library=emu_lib), a)
    assert R.h5.read(fn, "extra", library=emu_lib)[0] == 7.0
    with H.File(fn, "r") as f:
        assert f.paths() == ["extra", "g/a", "g/b", "time"]


def test_h5lite_appends_to_a_libhdf5_file(emu_lib, tmp_path):
    """The other direction: a file created by the library, extended by h5lite, read back by the library."""
    fn = str(tmp_path / "mixed2.h5")
    a = np.linspace(0.0, 1.0, 35).reshape(5, 7)
    with H.File(fn, "w") as f:
        f.write("g/a", a)
        f.write("time", np.array([1.5]))
    R.h5.write(fn, "g/c", 3.0 * a, library=emu_lib)
    R.h5.write(fn, "h/d", a.T.copy(), library=emu_lib)
    with H.File(fn, "r") as f:
        got = f.datasets()
    assert sorted(got) == ["g/a", "g/c", "h/d", "time"]
    assert np.array_equal(got["g/a"], a) and np.array_equal(got["g/c"], 3.0 * a) and np.array_equal(got["h/d"], a.T)


def test_h5dump_accepts_a_snapshot(emu_lib, tmp_path):
    """The library's own dump tool walks the whole file (every object header, heap and B-tree h5lite wrote)."""
    h5dump = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if os.path.exists("/opt/conda/bin/h5dump") else None)
    if h5dump is None:
        pytest.skip("no h5dump on this box")
    nav, _ = K.make_pair(emu_lib, False, 17, 17, 1e4, 1.0, 0.01, 1.0)
    nav.update(2)
    fn = str(tmp_path / "flow.h5")
    nav.write(fn)
    r = subprocess.run([h5dump, "-H", fn], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr[-2000:]
    for g, _ in GROUPS:
        assert f'GROUP "{g}"' in r.stdout, r.stdout[:2000]
    assert r.stdout.count("DATASET") == len(R.h5.paths(fn, library=emu_lib))
    r = subprocess.run([h5dump, "-d", "/time", fn], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "H5T_IEEE_F64LE" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("periodic", [False, True])
def test_libhdf5_reads_the_other_files_the_engines_write(emu_lib, tmp_path, periodic):
    """statistics.h5 (statistics.rs:116-161, with its unsigned 64-bit `num_save`), the snapshot of the adjoint-descent solver
    (steady_adjoint_io.rs) and the gradient file of Navier2DLnse::grad_adjoint: every dataset as libhdf5 sees it equals what the
    independent pure-Python parser (tests/h5classic.py) and h5lite's own reader see."""
    from tests.h5classic import File as Classic
    st, _ = K.check_statistics(emu_lib, periodic, 16 if periodic else 17, 17)
    files = [str(tmp_path / "statistics.h5")]
    st.write(files[0])
    mk = R.Navier2DAdjoint.new_periodic if periodic else R.Navier2DAdjoint.new_confined
    adj = mk(16 if periodic else 17, 17, 1e4, 1.0, 0.005, 1.0, "rbc", library=emu_lib)
    adj.set_velocity(0.1, 1.0, 1.0); adj.set_temperature(0.1, 1.0, 1.0)
    adj.update(2)
    files.append(str(tmp_path / "adjoint.h5"))
    adj.write(files[1])
    mkl = R.Navier2DLnse.new_periodic if periodic else R.Navier2DLnse.new_confined
    lin = mkl(16 if periodic else 17, 17, 3e3, 0.1, 0.01, 1.0, "rbc", library=emu_lib)
    os.makedirs(tmp_path / "data", exist_ok=True)
    files.append(str(tmp_path / "data" / "grad_adjoint.h5"))
    lin.grad_adjoint(0.05, None, 0.5, 0.5, None, filename=files[2])
    for fn in files:
        with H.File(fn, "r") as f:
            got = f.datasets()
        ref = Classic(fn).datasets
        assert sorted(got) == sorted(ref), fn
        for path, arr in got.items():
            assert np.array_equal(arr, np.asarray(ref[path]).reshape(arr.shape)), (fn, path)
    with H.File(files[0], "r") as f:
        assert f.read("num_save").dtype == np.uint64 and int(f.read("num_save")[0]) == st.num_save
