"""Statistics (src/navier_stokes/statistics.rs): device-side running mean / last-snapshot fields and the
Nusselt field against the oracle's restatement, the statistics.h5 layout through the independent parser, restart of the
statistics, and the callback hook (navier_io.rs:105-121)."""
import os

import numpy as np
import pytest

import rustpde_mpi_amd as R
from tests import checks as K
from tests.h5classic import File


@pytest.mark.parametrize("periodic,nx,ny", [(False, 17, 17), (False, 33, 17), (True, 16, 17)])
def test_statistics_match_the_oracle(emu_lib, periodic, nx, ny):
    K.check_statistics(emu_lib, periodic, nx, ny)


def test_statistics_time_mismatch_is_ignored(emu_lib, capfd):
    """statistics.rs:87-93: an update at a time before the statistics' own clock prints and returns."""
    nav, _ = K.make_pair(emu_lib, False, 17, 17, 1e4, 1.0, 0.01, 1.0)
    nav.update(3)
    st = R.Statistics.new(nav, 0.02, 0.04)     # tot_time = 0.03
    nav.reset_time()
    st.update()
    assert st.num_save == 0 and "Statistics time mismatch (navier < stat)" in capfd.readouterr().out


@pytest.mark.parametrize("periodic", [False, True])
def test_statistics_file_layout_and_restart(emu_lib, tmp_path, periodic):
    """Statistics::write / read (statistics.rs:116-161): groups temp, ux, uy, nusselt written like a Field2
    (field/io.rs:95-103), tot_time, avg_time, num_save (a usize: unsigned 64-bit) and the params."""
    st, so = K.check_statistics(emu_lib, periodic, 16 if periodic else 17, 17)
    nav = st._nav
    fn = str(tmp_path / "statistics.h5")
    st.write(fn)
    f = File(fn).datasets
    spec = ("vhat_re", "vhat_im") if periodic else ("vhat",)
    assert sorted(f) == sorted([f"{g}/{d}" for g in ("temp", "ux", "uy", "nusselt") for d in ("x", "dx", "y", "dy", "v") + spec] +
                               ["tot_time", "avg_time", "num_save", "ra", "pr", "nu", "ka"])
    assert f["num_save"].dtype == np.dtype("<u8") and f["num_save"][0] == 3
    assert f["tot_time"][0] == st.tot_time and f["avg_time"][0] == st.avg_time and f["ka"][0] == nav.params["ka"]
    for g, mine, theirs in (("temp", st.t_avg, so.t_avg), ("ux", st.ux_avg, so.ux_avg), ("uy", st.uy_avg, so.uy_avg),
                            ("nusselt", st.nusselt, so.nusselt)):
        vh = mine.vhat
        if periodic:
            assert np.array_equal(f[g + "/vhat_re"], vh.real) and np.array_equal(f[g + "/vhat_im"], vh.imag)
        else:
            assert np.array_equal(f[g + "/vhat"], vh)
        theirs.backward()                                  # Statistics::write calls backward() on every member first
        assert K.rel(f[g + "/v"], theirs.v) < 1e-11, g
        assert np.array_equal(f[g + "/x"], f[g + "/dx"]) and len(f[g + "/y"]) == 17
    # restart of the statistics into a fresh engine
    ctor = R.Navier2D.new_periodic if periodic else R.Navier2D.new_confined
    nav2 = ctor(nav.nx, nav.ny, 1e4, 1.0, 0.01, 1.0, "rbc", library=emu_lib)
    st2 = R.Statistics.new(nav2, 0.02, 0.04)
    st2.read(fn)
    assert st2.num_save == 3 and st2.tot_time == st.tot_time and st2.avg_time == st.avg_time
    for a, b in ((st.t_avg, st2.t_avg), (st.ux_avg, st2.ux_avg), (st.uy_avg, st2.uy_avg), (st.nusselt, st2.nusselt)):
        assert np.array_equal(a.vhat, b.vhat)


def test_callback_updates_and_writes_statistics(emu_lib, tmp_path, monkeypatch):
    """navier_io.rs:105-121: update on `save_stat`, data/statistics.h5 on `write_stat`."""
    monkeypatch.chdir(tmp_path)
    nav, _ = K.make_pair(emu_lib, False, 17, 17, 1e4, 1.0, 0.01, 1.0)
    nav.statistics = R.Statistics.new(nav, 0.02, 0.04)
    R.integrate(nav, 0.08, save_intervall=0.01)            # callbacks at t = 0.01 ... 0.08
    st = nav.statistics
    assert st.num_save == 4                                # t = 0.02, 0.04, 0.06, 0.08
    assert abs(st.tot_time - 0.08) < 1e-12 and abs(st.avg_time - 0.08) < 1e-12
    f = File("data/statistics.h5").datasets
    assert f["num_save"][0] == 4 and abs(f["tot_time"][0] - 0.08) < 1e-12
    with pytest.raises(R.RpdeError):
        other, _ = K.make_pair(emu_lib, False, 17, 17, 1e4, 1.0, 0.01, 1.0)
        other.statistics = st                              # statistics belong to their engine


def test_statistics_none_detaches_the_hook(emu_lib, tmp_path, monkeypatch):
    """`statistics: Option<Statistics>` (navier.rs:88): the callback acts on Some only (navier_io.rs:105); creating a
    Statistics does not hook it, assigning None takes the hook off and keeps the data, assigning it again resumes."""
    monkeypatch.chdir(tmp_path)
    nav, _ = K.make_pair(emu_lib, False, 17, 17, 1e4, 1.0, 0.01, 1.0)
    st = R.Statistics.new(nav, 0.02, 0.04)                 # not assigned: Statistics::new alone changes nothing
    R.integrate(nav, 0.04, save_intervall=0.01)
    assert st.num_save == 0 and not os.path.exists("data/statistics.h5")
    nav.statistics = st
    R.integrate(nav, 0.08, save_intervall=0.01)
    assert st.num_save == 2                                # t = 0.06, 0.08
    nav.statistics = None
    R.integrate(nav, 0.12, save_intervall=0.01)
    assert nav.statistics is None and st.num_save == 2     # detached: no update, the data stays
    nav.statistics = st
    R.integrate(nav, 0.14, save_intervall=0.01)
    assert st.num_save == 3


@pytest.mark.gpu
def test_statistics_on_the_gpu(hip_lib, tmp_path):
    st, so = K.check_statistics(hip_lib, False, 129, 65, ra=1e5)
    fn = str(tmp_path / "statistics.h5")
    st.write(fn)
    f = File(fn).datasets
    assert f["num_save"][0] == 3 and np.array_equal(f["nusselt/vhat"], st.nusselt.vhat)
    K.check_statistics(hip_lib, True, 64, 33)
