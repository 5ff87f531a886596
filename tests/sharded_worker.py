"""Worker of the world-size-N tests of the pencil-sharded engine (spawned by test_sharded.py or run
under torch.distributed.run).  Every rank builds the sharded engine, steps it, gathers the fields
through the C ABI and rank 0 compares them with the oracle and with the single-rank engine."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(rank, world, port, lib_path, device_build, cases, out_path):
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rustpde_mpi_amd as R
    from rustpde_mpi_amd._capi import Lib
    from rustpde_mpi_amd.dist import TorchComm
    from oracle import navier as N

    lib = Lib(lib_path)
    comm = TorchComm(device_buffers=device_build)
    results = []
    for case in cases:
        periodic, nx, ny, ra, dt, steps, aspect = case[:7]
        bc = case[7] if len(case) > 7 else "rbc"       # "hc": horizontal convection (three-term temperature base along y)
        vs_single = len(case) > 8 and case[8] == "single"   # compare with the one-rank engine (same setup code) instead of the oracle
        ab_overlap = len(case) > 8 and case[8] == "overlap_ab"   # the same run with RPDE_OVERLAP=0: fields must be bit-identical
        ctor = "new_periodic" if periodic else "new_confined"
        # the reference's spelling of the sharded constructors: Navier2DMpi::new_confined(&universe, nx, ny, ...)
        if ab_overlap:
            os.environ["RPDE_OVERLAP"] = "1"   # the default of the native RCCL transport, forced here for the callback transport
        try:
            nav = getattr(R.Navier2DMpi, ctor)(comm, nx, ny, ra, 1.0, dt, aspect, bc, library=lib)
        finally:
            os.environ.pop("RPDE_OVERLAP", None)
        nav.set_velocity(0.2, 1.0, 1.0)
        nav.set_temperature(0.2, 1.0, 1.0)
        nav.update(steps)
        got = nav.physical_fields()          # collective: every rank gathers the full fields
        got["pseu"] = nav.pseu.vhat
        divn = nav.div_norm()
        stats = nav.comm_stats()
        serial = None
        if ab_overlap:
            os.environ["RPDE_OVERLAP"] = "0"
            try:
                nav0 = getattr(R.Navier2DMpi, ctor)(comm, nx, ny, ra, 1.0, dt, aspect, bc, library=lib)
            finally:
                os.environ.pop("RPDE_OVERLAP", None)
            nav0.set_velocity(0.2, 1.0, 1.0)
            nav0.set_temperature(0.2, 1.0, 1.0)
            nav0.update(steps)
            f0 = nav0.physical_fields()
            f0["pseu"] = nav0.pseu.vhat
            serial = {"bitwise_equal": bool(all(np.array_equal(got[k], f0[k]) for k in f0)), "comm": nav0.comm_stats()}
            del nav0
        st = R.Statistics.new(nav, 1.0, 1.0)   # collective too: every rank gathers, reduces and keeps the same statistics
        st.update()
        got["stat_temp"], got["stat_nusselt"] = st.t_avg.vhat, st.nusselt.vhat
        if rank == 0 and (nx * ny > 1500 * 1500 or vs_single):
            # big grids: compare with the single-device engine instead of the (slow) oracle
            one = getattr(R.Navier2D, ctor)(nx, ny, ra, 1.0, dt, aspect, bc, library=lib)
            one.set_velocity(0.2, 1.0, 1.0)
            one.set_temperature(0.2, 1.0, 1.0)
            one.update(steps)
            want = one.physical_fields()
            want["pseu"] = one.pseu.vhat
            got.pop("stat_temp"); got.pop("stat_nusselt")   # statistics are compared with the oracle on the small cases
            err = {k: float(np.linalg.norm(got[k] - want[k]) / max(np.linalg.norm(want[k]), 1e-300)) for k in want}
            results.append({"case": [periodic, nx, ny, steps], "err": err, "div": [divn, one.div_norm()],
                            "comm": stats, "calls": comm.calls})
            del one
        elif rank == 0:
            ora = getattr(N.Navier2D, ctor)(nx, ny, ra, 1.0, dt, aspect, bc, eig_mode="parity")
            ora.set_velocity(0.2, 1.0, 1.0)
            ora.set_temperature(0.2, 1.0, 1.0)
            for _ in range(steps):
                ora.update()
            want = ora.physical_fields()
            want["pseu"] = ora.pseu.vhat
            so = N.Statistics(ora, 1.0, 1.0)
            so.update_from(ora)
            want["stat_temp"], want["stat_nusselt"] = so.t_avg.vhat, so.nusselt.vhat
            err = {k: float(np.linalg.norm(got[k] - want[k]) / max(np.linalg.norm(want[k]), 1e-300)) for k in want}
            results.append({"case": [periodic, nx, ny, steps], "err": err, "div": [divn, ora.div_norm()],
                            "comm": stats, "calls": comm.calls, "serial": serial})
        del nav
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump(results, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    # torch.distributed.run entry: python tests/sharded_worker.py <lib_path> <device_build 0|1> <out.json>
    cases = [(False, 33, 33, 1e5, 0.01, 5, 1.0), (True, 32, 33, 1e5, 0.01, 5, 1.0)]
    run(int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("MASTER_PORT", "29511")),
        sys.argv[1], bool(int(sys.argv[2])), cases, sys.argv[3])
