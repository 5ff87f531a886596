"""libhdf5-side check of the snapshot files (the consumers are plot/plot2d.py:30-54, which opens the
files with h5py, and src/io/read_write_hdf5.rs:38-188, which goes through the hdf5 crate = libhdf5).

csrc/h5lite writes and parses the classic HDF5 structures by hand; the checker of the other tests (tests/h5classic.py) is a
second reading of the same specification by the same author.  These tests use the real library through h5py and SKIP where
it is not installed (it is not in the build image -- tests/test_libhdf5_interop.py runs the same checks through a ctypes
binding to the image's libhdf5 and does not skip): on any box that has h5py they (i) open an h5lite snapshot with libhdf5 and compare every dataset, (ii) write the same layout with libhdf5's
defaults and restart an engine from it, (iii) let libhdf5 append to an h5lite file and read the result back through h5lite."""
import numpy as np
import pytest

import rustpde_mpi_amd as R
from tests import checks as K

h5py = pytest.importorskip("h5py", reason="no h5py / libhdf5 on this box")

GROUPS = (("ux", "velx"), ("uy", "vely"), ("temp", "temp"), ("pres", "pres"))


def _collect(h5file):
    out = {}
    h5file.visititems(lambda name, obj: out.__setitem__(name, np.asarray(obj)) if isinstance(obj, h5py.Dataset) else None)
    return out


@pytest.mark.parametrize("periodic", [False, True])
def test_libhdf5_reads_an_h5lite_snapshot(emu_lib, tmp_path, periodic):
    nav, _ = K.make_pair(emu_lib, periodic, 16 if periodic else 17, 17, 1e4, 1.0, 0.01, 1.0)
    nav.update(3)
    fn = str(tmp_path / "flow.h5")
    nav.write(fn)
    with h5py.File(fn, "r") as f:                              # what plot2d.py does
        got = _collect(f)
    assert sorted(got) == sorted(R.h5.paths(fn, library=emu_lib))
    for path, arr in got.items():
        assert arr.dtype == np.float64
        assert np.array_equal(arr, R.h5.read(fn, path, library=emu_lib)), path
    x, y = nav.velx.x
    for g, name in GROUPS:
        assert np.array_equal(got[g + "/v"], getattr(nav, name).v)
        assert np.array_equal(got[g + "/x"], x) and np.array_equal(got[g + "/y"], y)
    assert got["time"].shape == (1,) and got["time"][0] == nav.get_time()


@pytest.mark.parametrize("periodic", [False, True])
def test_engine_restarts_from_a_libhdf5_written_snapshot(emu_lib, tmp_path, periodic):
    """The same layout written by libhdf5 with its defaults (contiguous datasets, symbol-table groups -- what the
    reference's hdf5 crate produces): Navier2D::read must restore the state bit for bit."""
    nav, _ = K.make_pair(emu_lib, periodic, 16 if periodic else 17, 17, 1e4, 1.0, 0.01, 1.0)
    nav.update(3)
    fn = str(tmp_path / "by_libhdf5.h5")
    x, y = nav.velx.x
    with h5py.File(fn, "w", libver="earliest") as f:
        for g, name in GROUPS:
            fld = getattr(nav, name)
            grp = f.create_group(g)
            for k, v in (("x", x), ("dx", x), ("y", y), ("dy", y), ("v", fld.v)):
                grp.create_dataset(k, data=v)
            vh = fld.vhat
            if periodic:
                grp.create_dataset("vhat_re", data=np.ascontiguousarray(vh.real))
                grp.create_dataset("vhat_im", data=np.ascontiguousarray(vh.imag))
            else:
                grp.create_dataset("vhat", data=vh)
        for k, v in (("time", nav.get_time()), ("ra", 1e4), ("pr", 1.0), ("nu", nav.params["nu"]), ("ka", nav.params["ka"])):
            f.create_dataset(k, data=np.array([v]))
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
    with h5py.File(fn, "a") as f:
        f["g"].create_dataset("b", data=2.0 * a)
        f.create_dataset("extra", data=np.array([7.0]))
    assert np.array_equal(R.h5.read(fn, "g/b", library=emu_lib), 2.0 * a)
    assert np.array_equal(R.h5.read(fn, "g/a", library=emu_lib), a)
    assert R.h5.read(fn, "extra", library=emu_lib)[0] == 7.0
