"""The N > 1 plumbing on CPU: two processes, gloo backend.  Checks the hand-off collectives
(counts + variable-size record exchange), the completion all-reduce, and the global stream-id
bookkeeping of SourcePhotons across ranks."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from jaybenne_amd.comm import RECORD_WORDS, Comm
        comm = Comm()
        assert comm.device.type == "cpu" and comm.nranks == world
        # 1. record exchange: rank r sends (r + 1) * 3 records to the other rank
        n = (rank + 1) * 3
        send = torch.arange(n * RECORD_WORDS, dtype=torch.int64).reshape(n, RECORD_WORDS) + 1000 * rank
        counts = np.zeros(world, dtype=np.int64)
        counts[1 - rank] = n
        recv = comm.exchange_records(send, counts, torch.device("cpu"))
        other = 1 - rank
        m = (other + 1) * 3
        want = torch.arange(m * RECORD_WORDS, dtype=torch.int64).reshape(m, RECORD_WORDS) + 1000 * other
        assert torch.equal(recv, want)
        # 2. nothing to send on one side, nothing at all on the next iteration
        counts = np.zeros(world, dtype=np.int64)
        s2 = None
        if rank == 0:
            counts[1] = 2
            s2 = send[:2]
        r2 = comm.exchange_records(s2, counts, torch.device("cpu"))
        assert (r2 is None) if rank == 0 else (r2.shape[0] == 2)
        assert comm.exchange_records(None, np.zeros(world, dtype=np.int64), torch.device("cpu")) is None
        # 2b. the count matrix (one all-gather answers "how much do I receive" and "did anything
        # move anywhere")
        mine = np.zeros(world, dtype=np.int64)
        mine[1 - rank] = 5 + rank
        mat = comm.gather_count_matrix(mine)
        assert mat.tolist() == [[0, 5], [6, 0]]
        s3 = torch.zeros((5 + rank, RECORD_WORDS), dtype=torch.int64)
        r3 = comm.exchange_records(s3, mine, torch.device("cpu"), recv_counts=mat[:, rank].copy())
        assert r3.shape[0] == (6 if rank == 0 else 5)
        # 3. completion test: sum of arrivals
        tot = comm.allreduce_sum_int64(np.array([0 if rank == 0 else 2], dtype=np.int64))
        assert tot[0] == 2
        # 4. global per-block counts -> identical id bases on every rank
        from helpers import load_deck
        from jaybenne_amd.mesh import Mesh
        mesh = Mesh.from_deck(load_deck("stepdiff_smr"))
        owner = mesh.partition(world)
        local = np.nonzero(owner == rank)[0]
        counts = np.zeros(mesh.nblocks, dtype=np.int64)
        counts[local] = 100 + local
        allc = comm.allreduce_sum_int64(counts)
        assert np.array_equal(allc, 100 + np.arange(mesh.nblocks))
        assert comm.allreduce_max_float(float(rank)) == 1.0
        # a rank never sends to itself
        with pytest.raises(ValueError):
            bad = np.zeros(world, dtype=np.int64)
            bad[rank] = 1
            comm.exchange_records(send[:1], bad, torch.device("cpu"))
        comm.barrier()
        out.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        out.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_handoff():
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(out.get(timeout=5) for _ in range(2))
    assert res == {0: "ok", 1: "ok"}
