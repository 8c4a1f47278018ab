"""The collectives of jaybenne_amd/comm.py on the production backend (RCCL, ``"nccl"``), as far as
one GPU allows: a one-rank process group.  Checks that every call shape the multi-rank path uses
is accepted by RCCL for these dtypes (int64 / float64 all-to-all-v with empty and non-empty
splits, flat all-gather, all-reduce) and that a driver handed a communicator steps correctly.
Multi-rank semantics are covered with gloo (tests/test_comm_gloo.py, tests/test_gpu_multirank.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(port, out):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    try:
        from helpers import load_deck
        from jaybenne_amd import mcblock
        from jaybenne_amd.comm import RECORD_WORDS, Comm
        comm = Comm(device=device)
        assert comm.device.type == "cuda" and comm.nranks == 1
        assert comm.allreduce_sum_int64(np.array([3, 4], dtype=np.int64)).tolist() == [3, 4]
        assert comm.allreduce_max_float(2.5) == 2.5
        assert comm.gather_count_matrix(np.array([0], dtype=np.int64)).tolist() == [[0]]
        assert comm.exchange_counts(np.array([7], dtype=np.int64)).tolist() == [7]
        # nothing to send and nothing to receive: the all-to-all-v with empty splits
        assert comm.exchange_records(None, np.zeros(1, dtype=np.int64), device) is None
        empty = torch.empty((0, RECORD_WORDS), dtype=torch.int64, device=device)
        assert comm.exchange_records(empty, np.zeros(1, dtype=np.int64), device,
                                     recv_counts=np.zeros(1, dtype=np.int64)) is None
        # float64 / int64 payloads (a rank may address values to itself in exchange_values)
        send = torch.arange(5, dtype=torch.float64, device=device) + 0.5
        recv = torch.empty(5, dtype=torch.float64, device=device)
        comm.exchange_values(send, np.array([5]), recv, np.array([5]))
        assert torch.equal(send, recv)
        got = comm.exchange_int64_lists([np.arange(4, dtype=np.int64)])
        assert got[0].tolist() == [0, 1, 2, 3]
        comm.barrier()
        # a driver with a communicator (one rank) equals a driver without
        ov = {"jaybenne/num_particles": 4000}
        a = mcblock.McblockDriver(load_deck("stepdiff", ov), rank=0, nranks=1, comm=comm, device=device)
        b = mcblock.McblockDriver(load_deck("stepdiff", ov), device=device)
        for d in (a, b):
            d.Step()
        ga, gb = a.md.get_swarm(), b.md.get_swarm()
        for k in ga:
            assert np.array_equal(ga[k], gb[k]), k
        out.put("ok")
    except Exception as e:  # pragma: no cover
        out.put(repr(e))
        raise
    finally:
        dist.destroy_process_group()


def test_rccl_one_rank_group(gpu_device):
    sys.path.insert(0, os.path.dirname(__file__))
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), out))
    p.start()
    p.join(300)
    assert p.exitcode == 0
    assert out.get(timeout=5) == "ok"


def _mpi_env():
    import shutil
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mpiexec = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"
    if not os.path.exists(mpiexec) or not os.path.exists("/opt/conda/lib/libmpi.so"):
        pytest.skip("no MPI installation in this image")
    import subprocess
    build = subprocess.run(["make", "-C", os.path.join(root, "examples"), "mpi"], capture_output=True, text=True)
    assert build.returncode == 0, build.stdout + build.stderr
    return root, mpiexec


def test_c_level_exchange_over_rccl_one_rank(gpu_device, tmp_path):
    """jb_exchange over jb_transport_rccl (include/jaybenne_amd.h), from C++ with no Python in the process
    (examples/handoff_mpi.cpp, exchange = rccl): the program makes a one-rank RCCL communicator
    (ncclGetUniqueId / ncclCommInitRank -- one card here, so one rank), sends five records to itself through
    the transport's grouped ncclSend / ncclRecv, and runs three cycles whose every transport iteration goes
    through jb_exchange (device-side count -> ncclAllGather -> one read-back: nothing moved) -- the photons
    equal the oracle's bit for bit."""
    import subprocess
    from helpers import load_deck, make_oracle, run_oracle_cycles
    from oracle import orc
    from test_gpu_multirank import _read_photon_dumps
    root, mpiexec = _mpi_env()
    ov = {"parthenon/mesh/nx1": 128, "parthenon/meshblock/nx1": 16, "jaybenne/num_particles": 50000}
    O, mesh, _ = make_oracle(load_deck("stepdiff", ov), orc.MATH_PORTABLE)
    run_oracle_cycles(O, load_deck("stepdiff", ov), 3)
    order = np.argsort(O.sw["id"][:O.n])
    prefix = str(tmp_path / "photons")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    run = subprocess.run([mpiexec, "-n", "1", os.path.join(root, "examples", "handoff_mpi"), "16", "8", "50000", "3",
                          "1", prefix, "rccl"], capture_output=True, text=True, timeout=240, env=env)
    assert run.returncode == 0 and "HANDOFF OK" in run.stdout, run.stdout + run.stderr
    assert "RCCL transport: grouped self send / recv of 5 records ok" in run.stdout
    g = _read_photon_dumps(prefix, 1)
    assert len(g) == O.n and np.array_equal(g["id"], O.sw["id"][:O.n][order])
    for k in ("x", "vx", "t", "w", "rng", "ip"):
        assert np.array_equal(g[k], O.sw[k][:O.n][order]), k


@pytest.mark.parametrize("rings", [1, 0])
def test_c_level_exchange_three_ranks_over_the_mpi_transport(gpu_device, tmp_path, rings):
    """The same jb_exchange -- counts on the device, rank x rank matrix, pack, all-to-all-v, unpack, the
    capacity protocol -- on three ranks that share the card, over a jb_exchange_transport written with MPI
    (RCCL wants one GPU per rank): with halo copies two transport iterations per cycle, without them
    dozens; either way the union of the ranks' photons equals the single-process oracle bit for bit."""
    import re
    import subprocess
    from helpers import load_deck, make_oracle, run_oracle_cycles
    from oracle import orc
    from test_gpu_multirank import _read_photon_dumps
    root, mpiexec = _mpi_env()
    ov = {"parthenon/mesh/nx1": 128, "parthenon/meshblock/nx1": 16, "jaybenne/num_particles": 200000}
    O, mesh, _ = make_oracle(load_deck("stepdiff", ov), orc.MATH_PORTABLE)
    run_oracle_cycles(O, load_deck("stepdiff", ov), 3)
    order = np.argsort(O.sw["id"][:O.n])
    prefix = str(tmp_path / f"photons{rings}")
    run = subprocess.run([mpiexec, "-n", "3", os.path.join(root, "examples", "handoff_mpi"), "16", "8", "200000", "3",
                          str(rings), prefix, "mpi"], capture_output=True, text=True, timeout=240)
    assert run.returncode == 0 and "HANDOFF OK" in run.stdout, run.stdout + run.stderr
    per_cycle = float(re.search(r"\(([0-9.]+) per cycle\)", run.stdout).group(1))
    assert per_cycle == 2.0 if rings == 1 else per_cycle > 20.0
    g = _read_photon_dumps(prefix, 3)
    assert len(g) == O.n and np.array_equal(g["id"], O.sw["id"][:O.n][order])
    for k in ("x", "vx", "t", "w", "rng", "ip"):
        assert np.array_equal(g[k], O.sw[k][:O.n][order]), (rings, k)
    assert np.array_equal(g["gblk"], O.sw["blk"][:O.n][order])


def test_c_level_exchange_capacity_protocol_is_collective(gpu_device, tmp_path):
    """jb_exchange's JB_ERR_CAPACITY comes out on EVERY rank in the same call (each rank's room travels with
    its counts in the all-gather), so no rank is left waiting in the payload exchange: (i) swarms with little
    room -- the three ranks close their holes and go again, several times per run, and the photons still
    equal the oracle's; (ii) a record buffer of ten records -- all three ranks stop with the message, at once,
    instead of two of them hanging until the timeout."""
    import re
    import subprocess
    from helpers import load_deck, make_oracle, run_oracle_cycles
    from oracle import orc
    from test_gpu_multirank import _read_photon_dumps
    root, mpiexec = _mpi_env()
    ov = {"parthenon/mesh/nx1": 128, "parthenon/meshblock/nx1": 16, "jaybenne/num_particles": 200000}
    O, mesh, _ = make_oracle(load_deck("stepdiff", ov), orc.MATH_PORTABLE)
    run_oracle_cycles(O, load_deck("stepdiff", ov), 2)
    order = np.argsort(O.sw["id"][:O.n])
    prefix = str(tmp_path / "photons")
    exe = os.path.join(root, "examples", "handoff_mpi")
    # (i) 200000 photons on three ranks, room for 90000 each, no halo copies: ~75 iterations per cycle append
    # arrivals behind holes until the swarm is full
    env = dict(os.environ, JB_HANDOFF_CAPACITY="90000")
    run = subprocess.run([mpiexec, "-n", "3", exe, "16", "8", "200000", "2", "0", prefix, "mpi"],
                         capture_output=True, text=True, timeout=240, env=env)
    assert run.returncode == 0 and "HANDOFF OK" in run.stdout, run.stdout + run.stderr
    assert int(re.search(r"JB_ERR_CAPACITY (\d+) time", run.stdout).group(1)) >= 2, run.stdout
    g = _read_photon_dumps(prefix, 3)
    assert len(g) == O.n and np.array_equal(g["id"], O.sw["id"][:O.n][order])
    for k in ("x", "vx", "t", "w", "rng", "ip"):
        assert np.array_equal(g[k], O.sw[k][:O.n][order]), k
    # (ii) ten records of buffer: every rank reports, none hangs
    env = dict(os.environ, JB_HANDOFF_REC_CAP="10")
    run = subprocess.run([mpiexec, "-n", "3", exe, "16", "8", "200000", "1", "0", "-", "mpi"],
                         capture_output=True, text=True, timeout=120, env=env)
    assert run.returncode != 0
    assert "buffer holds 10 records" in run.stdout + run.stderr
