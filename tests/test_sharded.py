"""Pencil-sharded engine (the reference's Navier2DMpi path) with world size 2 and 3 over gloo.

CPU: the host emulation build of the kernel sources + gloo (runs in this container).
GPU: the HIP build with several ranks sharing the one GPU of the test box + gloo (host staging) --
this exercises the device pack / unpack / halo code; the RCCL transport itself is only the
`all_to_all_single` call in rustpde_mpi_amd/dist.py."""
import json
import os
import socket

import pytest


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn(world, lib_path, device_build, cases, tmp_path):
    import torch.multiprocessing as mp
    from tests.sharded_worker import run
    out = str(tmp_path / f"sharded_{world}.json")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    mp.spawn(run, args=(world, int(os.environ["MASTER_PORT"]), lib_path, device_build, cases, out),
             nprocs=world, join=True)
    with open(out) as f:
        return json.load(f)


CASES = [(False, 17, 17, 1e4, 0.01, 3, 1.0), (False, 33, 17, 1e5, 0.01, 6, 2.0),
         (True, 16, 17, 1e5, 0.01, 4, 1.0), (True, 32, 33, 1e5, 0.01, 6, 1.0)]
# several column-scan blocks per rank with a ragged last one (129 rows), and the line length the whole-line kernels
# of the emulation build cover (257): the kernels of the single-GPU step run on the local lines of every rank
# (256 reals per x-line: the Fourier whole-line kernels of the emulation build, rfft_line.h, on the local lines of a rank)
CASES_BLOCKS = [(False, 129, 129, 1e5, 0.01, 3, 1.0), (True, 128, 129, 1e5, 0.01, 3, 1.0), (False, 257, 257, 1e6, 0.005, 3, 1.0),
                (True, 256, 65, 1e5, 0.01, 3, 1.0)]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_matches_oracle_emulation(world, tmp_path, emu_lib):
    res = _spawn(world, emu_lib.path, False, CASES, tmp_path)
    assert len(res) == len(CASES)
    for r in res:
        for k, e in r["err"].items():
            assert e < 1e-10, (r["case"], k, e)
        assert abs(r["div"][0] - r["div"][1]) < 1e-9 * max(1.0, r["div"][1])
        assert r["comm"][1] > 0 and r["comm"][0] > 0   # exchanges per step, bytes per step


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_column_scans_and_whole_line_kernels_emulation(world, tmp_path, emu_lib):
    """The y-direction stages of a sharded step are the column scans of the single-GPU step: a rank reduces its blocks
    to one summary per column, the summaries travel in one small exchange, every rank derives its inflow (colscan.h).
    13 exchanges per confined step: T1, T2, T4b, T4c, five halo exchanges, four column-scan summaries; 13 per periodic step too
    (round 4: its y-correction is the column scan of the confined step -- two array transposes less, one summary more)."""
    res = _spawn(world, emu_lib.path, False, CASES_BLOCKS, tmp_path)
    assert len(res) == len(CASES_BLOCKS)
    for r in res:
        for k, e in r["err"].items():
            assert e < (1e-10 if k == "pseu" else 1e-12), (r["case"], k, e)
        assert r["comm"][1] == 13   # the serial order: the callback transport (gloo here) cannot overlap (RPDE_OVERLAP defaults to on for the engine's own RCCL communicator only: +4, T1 and T2 as three exchanges each)


CASES_OVERLAP = [(False, 33, 33, 1e5, 0.01, 4, 1.0, "rbc", "overlap_ab"), (True, 32, 33, 1e5, 0.01, 4, 1.0, "rbc", "overlap_ab"),
                 (False, 129, 129, 1e5, 0.01, 3, 1.0, "rbc", "overlap_ab"), (False, 33, 33, 1e5, 0.01, 4, 1.0, "hc", "overlap_ab")]


@pytest.mark.parametrize("world", [2, 3])
def test_overlap_order_equals_serial_order_emulation(world, tmp_path, emu_lib):
    """RPDE_OVERLAP (round 5): the transposes of one field leave as soon as that field's producer has run (three T1 and three
    T2 exchanges instead of one batch each, on a second stream in the HIP build) and the consumers wait for the exchange they
    read.  Same arithmetic, another order of launches: the fields must be BIT-identical to the serial order, and both meet
    the oracle.  17 exchanges per step instead of 13 (six array exchanges instead of two)."""
    res = _spawn(world, emu_lib.path, False, CASES_OVERLAP, tmp_path)
    assert len(res) == len(CASES_OVERLAP)
    for r in res:
        for k, e in r["err"].items():
            assert e < 1e-10, (r["case"], k, e)
        assert r["serial"]["bitwise_equal"], r["case"]
        # "rbc": T1 and T2 were one batch each (2 -> 6 exchanges); "hc": T1 already went out as two batches (3 -> 6)
        assert r["comm"][1] - r["serial"]["comm"][1] in (3, 4), (r["comm"], r["serial"]["comm"])


# "hc": the three-term stencil of the temperature reads two halo rows, its seven-diagonal Helmholtz solve along y goes
# through x-pencils (two more array exchanges per step)
CASES_HC = [(False, 33, 33, 1e5, 0.01, 5, 1.0, "hc"), (True, 32, 33, 1e5, 0.01, 5, 1.0, "hc"), (False, 17, 65, 1e5, 0.01, 3, 2.0, "hc"),
            (False, 257, 33, 1e5, 0.01, 3, 1.0, "hc")]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_hc_matches_oracle_emulation(world, tmp_path, emu_lib):
    res = _spawn(world, emu_lib.path, False, CASES_HC, tmp_path)
    assert len(res) == len(CASES_HC)
    for r in res:
        for k, e in r["err"].items():
            assert e < 1e-10, (r["case"], k, e)
        # the "rbc" count + T3 / T4 of the temperature + one more for T1 (the temperature arrays have ny rows instead of my:
        # two batches)
        assert r["comm"][1] == 16   # (19 with RPDE_OVERLAP=1)


# BASELINE configs[3] / [4] run on 8 GPUs: 4097 = 8 * 512 + 1 rows is a ragged 8-way partition with several column-scan
# blocks per rank.  The same shape in miniature: 513 = 8 * 64 + 1 (confined) and 512 x 257 (periodic, 257 = 8 * 32 + 1).
# The confined case is compared with the ONE-RANK engine (same setup code, same eigenbasis): against the oracle the first
# steps of a 513^2 run carry the start-up transient of two LAPACK eigenbases (1.5e-10 on the velocities at step 2, the
# one-rank engine shows the same figure) -- the periodic case has no eigenbasis and meets the oracle.
CASES_WORLD8 = [(False, 513, 513, 1e6, 0.005, 2, 1.0, "rbc", "single"), (True, 512, 257, 1e6, 0.005, 2, 1.0)]


def test_sharded_world_size_8_emulation(tmp_path, emu_lib):
    """Eight ranks (the node size of BASELINE configs 4 and 5) over gloo: ragged row / column partitions, two column-scan
    blocks per rank, the summaries of eight ranks in the column-scan exchange (kColMaxRanks)."""
    res = _spawn(8, emu_lib.path, False, CASES_WORLD8, tmp_path)
    assert len(res) == len(CASES_WORLD8)
    for r in res:
        for k, e in r["err"].items():
            assert e < (1e-10 if r["case"][0] else 1e-11), (r["case"], k, e)
        assert abs(r["div"][0] - r["div"][1]) < 1e-9 * max(1.0, r["div"][1])
        assert r["comm"][1] == 13


def test_sharded_long_fourier_lines_emulation(tmp_path, emu_lib):
    """BASELINE configs[4] geometry in miniature: periodic, x-lines of 8192 reals (the one-slot
    1024-thread line configuration), aspect 8, two ranks."""
    res = _spawn(2, emu_lib.path, False, [(True, 8192, 9, 1e5, 0.01, 2, 8.0)], tmp_path)
    for k, e in res[0]["err"].items():
        assert e < 1e-10, (k, e)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_sharded_matches_oracle_hip(world, tmp_path, hip_lib):
    cases = CASES + [(False, 129, 65, 1e5, 0.01, 10, 1.0), (True, 128, 65, 1e5, 0.01, 10, 1.0)] + CASES_HC[:3]
    res = _spawn(world, hip_lib.path, True, cases, tmp_path)
    for r in res:
        for k, e in r["err"].items():
            assert e < 1e-10, (r["case"], k, e)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_overlap_order_equals_serial_order_hip(world, tmp_path, hip_lib):
    """The same A/B on the device: the per-field exchanges run on the engine's second stream behind an event of their
    producer, the consumers wait for the exchange's event (ranks share the one GPU of the test box; gloo callback
    transport) -- bit-identical to RPDE_OVERLAP=0, and both meet the oracle.  Includes a 1025-wide case (whole-line kernels on
    the local lines, several column-scan blocks per rank)."""
    cases = CASES_OVERLAP + [(False, 1025, 257, 1e6, 1e-3, 3, 1.0, "rbc", "overlap_ab")]
    res = _spawn(world, hip_lib.path, True, cases, tmp_path)
    assert len(res) == len(cases)
    for r in res:
        for k, e in r["err"].items():
            assert e < 1e-10, (r["case"], k, e)
        assert r["serial"]["bitwise_equal"], r["case"]


@pytest.mark.gpu
@pytest.mark.parametrize("n,world,dt", [(2049, 4, 5e-4), (4097, 2, 2e-4)])
def test_config4_geometry_sharded_equals_single_device(tmp_path, hip_lib, n, world, dt):
    """BASELINE.json configs[3] (confined 4097 x 4097, pencil-sharded; also 2049 x 2049 over 4 ranks; the ranks share
    the one GPU of the test box): the sharded fields equal the single-device fields after 3 steps."""
    res = _spawn(world, hip_lib.path, True, [(False, n, n, 1e8, dt, 3, 1.0)], tmp_path)
    for k, e in res[0]["err"].items():
        # u, v, T, p: 1e-11.  Same kernels on both sides; the association of the column-scan carries (per rank, then
        # across the ranks) and the GEMM tile shapes differ; pseu is the raw output of the Poisson solve, which
        # amplifies such round-off (measured 1.4e-11)
        assert e < (1e-9 if k == "pseu" else 1e-11), (k, e)
    assert res[0]["comm"][1] == 13   # 4 array all-to-alls (T1, T2, T4b, T4c) + 5 halo exchanges + 4 column-scan summaries per step (callback transport: serial order)


def _nccl_single(rank, port, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from rustpde_mpi_amd.dist import TorchComm
    comm = TorchComm(device_buffers=True)
    a = torch.arange(1000, dtype=torch.float64, device="cuda") * 0.5
    b = torch.zeros(1000, dtype=torch.float64, device="cuda")
    # the engine hands over raw device pointers: go through the C callback exactly as it does
    import ctypes as C
    sc = (C.c_int64 * 1)(1000)
    rc = (C.c_int64 * 1)(1000)
    rcode = comm.c_callback(None, a.data_ptr(), sc, b.data_ptr(), rc)
    ok = rcode == 0 and bool(torch.equal(a, b))
    with open(out, "w") as f:
        f.write("ok" if ok else "bad")
    dist.destroy_process_group()


@pytest.mark.gpu
def test_torchcomm_over_rccl_single_rank(tmp_path, hip_lib):
    """The RCCL transport of bench.py --gpus N: raw device pointers -> torch tensors ->
    all_to_all_single on backend nccl (world size 1 is all this box can offer)."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "nccl.txt")
    mp.spawn(_nccl_single, args=(_free_port(), out), nprocs=1, join=True)
    assert open(out).read() == "ok"


def _native_rccl_single(rank, out):
    import ctypes as C
    import numpy as np
    import torch
    import rustpde_mpi_amd as R
    L = R.lib()
    uid = C.create_string_buffer(128)
    L.call("rpde_rccl_unique_id", uid)
    torch.cuda.set_device(0)
    a = torch.arange(5000, dtype=torch.float64, device="cuda")
    b = torch.zeros(5000, dtype=torch.float64, device="cuda")
    sc = (C.c_int64 * 1)(5000)
    rc = (C.c_int64 * 1)(5000)
    torch.cuda.synchronize()
    L.call("rpde_rccl_alltoallv_once", uid.raw, 0, 1, 0, C.c_void_p(a.data_ptr()), C.cast(sc, C.c_void_p),
           C.c_void_p(b.data_ptr()), C.cast(rc, C.c_void_p))
    ok = bool(torch.equal(a, b))
    # the engine constructor with the native transport (communicator bound to the engine's device);
    # an ncclUniqueId serves exactly one communicator, so take a fresh one
    L.call("rpde_rccl_unique_id", uid)
    h = C.c_void_p()
    L.call("rpde_navier2d_create_sharded_rccl", 0, 33, 33, 1e4, 1.0, 1e-2, 1.0, b"rbc", 0, 0, 1, uid.raw, C.byref(h))
    nav = R.Navier2D(h, 33, 33, False, L)
    nav.set_velocity(0.2, 1.0, 1.0)
    nav.set_temperature(0.2, 1.0, 1.0)
    nav.update(3)
    ref = R.Navier2D.new_confined(33, 33, 1e4, 1.0, 1e-2, 1.0, "rbc")
    ref.set_velocity(0.2, 1.0, 1.0)
    ref.set_temperature(0.2, 1.0, 1.0)
    ref.update(3)
    ok = ok and bool(np.array_equal(nav.temp.v, ref.temp.v)) and not nav.exit()
    with open(out, "w") as f:
        f.write("ok" if ok else "bad")


@pytest.mark.gpu
def test_native_rccl_transport_single_rank(tmp_path, hip_lib):
    """The native transport of bench.py --gpus N (grouped ncclSend/ncclRecv on the engine's stream):
    communicator creation, a self all-to-all of device buffers, and an engine built on it.  World size
    1 is all a one-GPU box can offer (RCCL refuses two ranks on one device); the exchange schedule and
    counts are the ones the gloo tests above verify."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "rccl_native.txt")
    mp.spawn(_native_rccl_single, args=(out,), nprocs=1, join=True)
    assert open(out).read() == "ok"


def _native_rccl_two_ranks(rank, port, out):
    """One process per GPU; two engines from ONE RcclComm (each takes a fresh ncclUniqueId)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=2, device_id=torch.device("cuda", rank))
    import rustpde_mpi_amd as R
    from rustpde_mpi_amd.dist import RcclComm
    comm = RcclComm()
    errs = {}
    for periodic, nx, ny in ((False, 129, 65), (True, 128, 65)):
        ctor = R.Navier2D.new_periodic if periodic else R.Navier2D.new_confined
        nav = ctor(nx, ny, 1e5, 1.0, 0.01, 1.0, "rbc", device=rank, comm=comm)
        nav.set_velocity(0.2, 1.0, 1.0)
        nav.set_temperature(0.2, 1.0, 1.0)
        nav.update(5)
        bad = nav.exit()
        got = nav.physical_fields()
        if rank == 0:
            one = ctor(nx, ny, 1e5, 1.0, 0.01, 1.0, "rbc", device=0)
            one.set_velocity(0.2, 1.0, 1.0)
            one.set_temperature(0.2, 1.0, 1.0)
            one.update(5)
            want = one.physical_fields()
            for k in want:
                errs[f"{'p' if periodic else 'c'}:{k}"] = float(np.linalg.norm(got[k] - want[k]) / np.linalg.norm(want[k]))
            errs[f"{'p' if periodic else 'c'}:exit"] = float(bad)
        del nav
    if rank == 0:
        with open(out, "w") as f:
            json.dump(errs, f)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_native_rccl_two_ranks_two_gpus(tmp_path, hip_lib):
    """The native RCCL transport (grouped ncclSend/ncclRecv over xGMI on the engine's stream) with a
    REAL second rank: needs two GPUs, skips on a one-GPU box.  Also builds two engines from one
    RcclComm object (an ncclUniqueId serves exactly one ncclCommInitRank)."""
    import ctypes as C
    n = C.c_int()
    hip_lib.call("rpde_device_count", C.byref(n))
    if n.value < 2:
        pytest.skip(f"needs 2 GPUs for two RCCL ranks, this box has {n.value}")
    import torch.multiprocessing as mp
    out = str(tmp_path / "rccl2.json")
    mp.spawn(_native_rccl_two_ranks, args=(_free_port(), out), nprocs=2, join=True)
    errs = json.load(open(out))
    for k, e in errs.items():
        assert e < (1e-11 if not k.endswith("exit") else 0.5), (k, e)
