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
        # 5. replicated mesh, split particles: the ranks' shares of every block's new photons tile the
        # block exactly, and the one exchange of a cycle -- the sum of the ranks' cell fields -- leaves
        # the same bits on every rank
        from jaybenne_amd.jaybenne import rank_share
        nper = np.array([1000003, 0, 7, 64], dtype=np.int32)
        first, cnt = rank_share(nper, rank, world)
        ends = comm.allreduce_sum_int64(cnt.astype(np.int64))
        assert np.array_equal(ends, nper)
        assert np.array_equal(first, (nper.astype(np.int64) * rank) // world)
        field = torch.full((3, 5), 0.1 * (rank + 1), dtype=torch.float64)
        comm.allreduce_sum_tensor(field)
        assert torch.equal(field, torch.full((3, 5), 0.1, dtype=torch.float64) + torch.full((3, 5), 0.2, dtype=torch.float64))
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


# ------------------------------------------------------------------------------------------------
class _StubMeshData:
    """The attributes halo.FieldExchange reads from a MeshData, without a device."""

    def __init__(self, mesh, rank, nranks, comm, halo_rings=1):
        owner = np.asarray(mesh.owner)
        self.mesh, self.rank, self.nranks, self.comm = mesh, rank, nranks, comm
        self.device = torch.device("cpu")
        self.gids = np.nonzero(owner == rank)[0].astype(np.int32)
        halo = mesh.neighbours(self.gids, halo_rings) if nranks > 1 else np.zeros(0, dtype=np.int32)
        self.resident_gids = np.concatenate([self.gids, halo]).astype(np.int32)
        self.owned_flags = np.concatenate([np.ones(len(self.gids), dtype=np.int32),
                                           np.zeros(len(halo), dtype=np.int32)])
        self.local_index = np.full(mesh.nblocks, -1, dtype=np.int32)
        self.local_index[self.resident_gids] = np.arange(len(self.resident_gids), dtype=np.int32)


HALO_CASES = [("stepdiff", {"parthenon/mesh/nx1": 16, "parthenon/meshblock/nx1": 4}),   # 1-D, reflecting
              ("stepdiff_smr", {"parthenon/mesh/nx1": 32, "parthenon/mesh/nx2": 16,
                                "parthenon/meshblock/nx1": 8, "parthenon/meshblock/nx2": 8}),  # 2-D, 2 levels
              ("inf", {"parthenon/mesh/nx1": 8, "parthenon/mesh/nx2": 8, "parthenon/mesh/nx3": 8,
                       "parthenon/meshblock/nx1": 4, "parthenon/meshblock/nx2": 4,
                       "parthenon/meshblock/nx3": 4})]                                  # 3-D, periodic


def _halo_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from helpers import load_deck
        from jaybenne_amd.comm import Comm
        from jaybenne_amd.halo import FieldExchange
        from jaybenne_amd.mesh import Mesh
        comm = Comm()
        for deck, ov in HALO_CASES:
            mesh = Mesh.from_deck(load_deck(deck, ov))
            mesh.partition(world)
            md = _StubMeshData(mesh, rank, world, comm)
            ex = FieldExchange(md)
            # the truth: a global field with ghost zones filled by the single-process routine
            rng = np.random.default_rng(7)
            glob = rng.random(mesh.field_shape)
            want = glob.copy()
            mesh.fill_ghosts(want)
            # this rank's copy: owned interiors only, everything else poisoned
            mine = np.full((len(md.resident_gids),) + mesh.field_shape[1:], np.nan)
            sl = mesh.interior()
            nown = len(md.gids)
            mine[:nown][sl] = glob[md.gids][sl]
            # what refresh() does, with numpy standing in for the two kernels
            flat = mine.reshape(mine.shape[0], -1)
            send = torch.from_numpy(flat[ex.serve_blk.numpy(), ex.serve_cell.numpy()].copy())
            assert not torch.isnan(send).any()            # only owned interior cells are served
            remote = torch.empty(ex.nremote, dtype=torch.float64)
            comm.exchange_values(send, ex.send_counts, remote, ex.recv_counts)
            ex.refresh_numpy(mine, remote.numpy())
            got_ok = np.array_equal(mine, want[md.resident_gids])
            assert got_ok, (deck, rank, np.argwhere(mine != want[md.resident_gids])[:5])
            assert ex.nremote > 0 and len(md.resident_gids) > nown
        out.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        out.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


def test_two_rank_field_halo_exchange():
    """Ghost zones and halo copies refreshed across two ranks equal the single-process ghost
    fill bit for bit (1-D reflecting, 2-D two-level SMR, 3-D periodic)."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_halo_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    res = dict(out.get(timeout=5) for _ in range(2))
    assert res == {0: "ok", 1: "ok"}


# ------------------------------------------------------------------------------------------------
def _eight_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from helpers import load_deck
        from jaybenne_amd.comm import RECORD_WORDS, Comm
        from jaybenne_amd.halo import FieldExchange
        from jaybenne_amd.mesh import Mesh
        comm = Comm()
        # hand-off: rank s sends (s * world + d) % 5 records to every other rank d; each record's
        # first word names (source, destination, ordinal)
        counts = np.array([(rank * world + d) % 5 if d != rank else 0 for d in range(world)], dtype=np.int64)
        rows = []
        for d in range(world):
            for q in range(counts[d]):
                rows.append([rank * 10000 + d * 100 + q] + [0] * (RECORD_WORDS - 1))
        send = torch.tensor(rows, dtype=torch.int64).reshape(-1, RECORD_WORDS) if rows else None
        mat = comm.gather_count_matrix(counts)
        want_mat = np.array([[(s_ * world + d) % 5 if d != s_ else 0 for d in range(world)]
                             for s_ in range(world)])
        assert np.array_equal(mat, want_mat)
        recv = comm.exchange_records(send, counts, torch.device("cpu"), recv_counts=mat[:, rank].copy())
        want = [s_ * 10000 + rank * 100 + q for s_ in range(world) for q in range(want_mat[s_, rank])]
        got = [] if recv is None else recv[:, 0].tolist()
        assert got == want                               # grouped by source rank, in send order
        # termination: nothing moves -> the matrix is all zero on every rank
        assert comm.gather_count_matrix(np.zeros(world, dtype=np.int64)).sum() == 0
        # field halo refresh on the two-level SMR deck (20 blocks over 8 ranks: 2 or 3 each)
        mesh = Mesh.from_deck(load_deck("stepdiff_smr"))
        mesh.partition(world)
        md = _StubMeshData(mesh, rank, world, comm)
        ex = FieldExchange(md)
        rng = np.random.default_rng(11)
        glob = rng.random(mesh.field_shape)
        want_f = glob.copy()
        mesh.fill_ghosts(want_f)
        mine = np.full((len(md.resident_gids),) + mesh.field_shape[1:], np.nan)
        sl = mesh.interior()
        nown = len(md.gids)
        mine[:nown][sl] = glob[md.gids][sl]
        flat = mine.reshape(mine.shape[0], -1)
        sendv = torch.from_numpy(flat[ex.serve_blk.numpy(), ex.serve_cell.numpy()].copy())
        remote = torch.empty(ex.nremote, dtype=torch.float64)
        comm.exchange_values(sendv, ex.send_counts, remote, ex.recv_counts)
        ex.refresh_numpy(mine, remote.numpy())
        assert np.array_equal(mine, want_f[md.resident_gids])
        out.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        out.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


def test_eight_rank_gloo_handoff_and_halo():
    """The rank count of the north-star node (and of the reference's CI, ci.yml:129-140): 8 x 8
    count matrix, every rank pair exchanging records, arrival order, the all-zero termination
    matrix, and the ghost / halo refresh plan of the SMR deck split eight ways."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eight_worker, args=(r, 8, port, out)) for r in range(8)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = dict(out.get(timeout=5) for _ in range(8))
    assert res == {r: "ok" for r in range(8)}
