"""Inter-rank exchange for the history loop: one process per GPU, ``torch.distributed`` over
RCCL (backend "nccl") on xGMI, or gloo on CPU for tests.

The reference's only communication on this path is (i) the neighbour hand-off of particles that
left a rank's blocks (``MeshSend`` / ``MeshReceive``, reference jaybenne.cpp:36-61) and (ii) the
global completion test of the iterate-sublist (``TQ::global_sync | TQ::completion``,
jaybenne.cpp:130-131).  Both are expressed here as at most two small collectives per transport
iteration (``jaybenne._exchange``):

* ``gather_count_matrix``: ONE all-gather of every rank's per-destination record counts -- the
  rank x rank matrix carries each rank's receive sizes and the global total that decides
  termination (the completion test needs no collective of its own);
* ``exchange_records`` (only if something moved): one all-to-all-v of fixed-size particle records
  (13 x 8 bytes), sized from the matrix.  xGMI is a full point-to-point mesh, so every rank pair
  moves its share over its own link concurrently.

``exchange_counts`` (an all-to-all of counts) is the fallback of ``exchange_records`` for callers that
have no matrix; ``allreduce_sum_int64`` serves the source (global per-block photon counts, one per
source call) and the bench's statistics.  With material feedback the halo copies' fields are
refreshed once per cycle (``halo.py``: one all-to-all-v of cell values); there is no collective
inside the tracking kernel.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

RECORD_WORDS = 13


class Comm:
    def __init__(self, group=None, device: Optional[torch.device] = None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.rank = dist.get_rank(group)
        self.nranks = dist.get_world_size(group)
        backend = dist.get_backend(group)
        if backend == "nccl":
            if device is None:
                device = torch.device("cuda", torch.cuda.current_device())
            self.device = device
        else:
            self.device = torch.device("cpu")

    # -- small host-side reductions
    def allreduce_sum_int64(self, values: np.ndarray) -> np.ndarray:
        t = torch.from_numpy(np.ascontiguousarray(values, dtype=np.int64)).to(self.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def allreduce_max_float(self, value: float) -> float:
        t = torch.tensor([value], dtype=torch.float64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def allreduce_sum_tensor(self, t: torch.Tensor) -> None:
        """Sum of a float64 device tensor over the ranks, in place, the same bits on every rank (RCCL on
        the tensor itself; gloo through a host copy).  The replicated-mesh mode's only exchange."""
        if t.device == self.device:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        else:
            h = t.to(self.device)
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)

    def barrier(self) -> None:
        dist.barrier(group=self.group)

    # -- particle hand-off
    def exchange_counts(self, send_counts: np.ndarray) -> np.ndarray:
        s = torch.from_numpy(np.ascontiguousarray(send_counts, dtype=np.int64)).to(self.device)
        r = torch.empty_like(s)
        dist.all_to_all_single(r, s, group=self.group)
        return r.cpu().numpy()

    def gather_count_matrix(self, send_counts: np.ndarray) -> np.ndarray:
        """counts[s, r] = records rank s sends to rank r, known to every rank after ONE
        all-gather: it carries both each rank's receive sizes and the global total that decides
        termination (the reference's separate completion all-reduce, jaybenne.cpp:130-131)."""
        s = torch.from_numpy(np.ascontiguousarray(send_counts, dtype=np.int64)).to(self.device)
        out = torch.empty(self.nranks * self.nranks, dtype=torch.int64, device=self.device)
        dist.all_gather_into_tensor(out, s, group=self.group)
        return out.cpu().numpy().reshape(self.nranks, self.nranks)

    def exchange_records(self, send: Optional[torch.Tensor], send_counts: np.ndarray,
                         out_device: torch.device, recv_counts: Optional[np.ndarray] = None
                         ) -> Optional[torch.Tensor]:
        """send: [sum(send_counts), 13] int64 ordered by destination rank (or None).  Returns the
        records addressed to this rank, [nrecv, 13] int64 on ``out_device`` (or None).
        recv_counts: column of the count matrix if the caller already gathered it."""
        send_counts = np.asarray(send_counts, dtype=np.int64)
        if send_counts[self.rank] != 0:
            raise ValueError("a rank does not hand particles to itself")
        if recv_counts is None:
            recv_counts = self.exchange_counts(send_counts)
        nrecv = int(recv_counts.sum())
        if send is None:
            send = torch.empty((0, RECORD_WORDS), dtype=torch.int64, device=self.device)
        if int(send.shape[0]) != int(send_counts.sum()):
            raise ValueError("record buffer does not match the send counts")
        send_c = send.to(self.device).contiguous()
        recv = torch.empty((nrecv, RECORD_WORDS), dtype=torch.int64, device=self.device)
        dist.all_to_all_single(recv, send_c, output_split_sizes=[int(c) for c in recv_counts],
                               input_split_sizes=[int(c) for c in send_counts], group=self.group)
        if nrecv == 0:
            return None
        return recv.to(out_device)

    # -- field halo exchange (jaybenne_amd/halo.py)
    def exchange_int64_lists(self, lists) -> list:
        """lists[r]: int64 array for rank r.  Returns what every rank addressed to this one, as
        a list indexed by source rank (set-up time only)."""
        counts = np.array([len(a) for a in lists], dtype=np.int64)
        recv_counts = self.exchange_counts(counts)
        send = torch.from_numpy(np.ascontiguousarray(np.concatenate(lists) if len(lists) else
                                                     np.zeros(0, dtype=np.int64))).to(self.device)
        recv = torch.empty(int(recv_counts.sum()), dtype=torch.int64, device=self.device)
        dist.all_to_all_single(recv, send, output_split_sizes=[int(c) for c in recv_counts],
                               input_split_sizes=[int(c) for c in counts], group=self.group)
        out = recv.cpu().numpy()
        bounds = np.concatenate(([0], np.cumsum(recv_counts)))
        return [out[bounds[r]:bounds[r + 1]] for r in range(self.nranks)]

    def exchange_values(self, send: torch.Tensor, send_counts: np.ndarray, recv: torch.Tensor,
                        recv_counts: np.ndarray) -> None:
        """All-to-all-v of float64 values with split sizes known to both sides (static plan);
        ``recv`` is filled in place (device tensors with RCCL, staged through the host with gloo)."""
        s = send.to(self.device).contiguous()
        r = recv if recv.device == self.device else torch.empty(recv.shape, dtype=recv.dtype,
                                                                 device=self.device)
        dist.all_to_all_single(r, s, output_split_sizes=[int(c) for c in recv_counts],
                               input_split_sizes=[int(c) for c in send_counts], group=self.group)
        if r is not recv:
            recv.copy_(r)

