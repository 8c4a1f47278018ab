"""Ghost-zone and halo refresh of the host's cell fields across ranks.

In the reference the host fields ``density`` / ``internal_energy`` carry ``Metadata::FillGhost``
(src/mcblock/mcblock.cpp:66-70) and ``McblockDriver::HostUpdateTasks`` runs Parthenon's
``AddBoundaryExchangeTasks`` on them after every radiation step (mcblock_driver.cpp:58-74):
``UpdateFluid`` changed ``internal_energy`` in the interior cells, and the next step's DDMC face
probabilities read material state one cell across each block face (jaybenne.cpp:354-372).

Here a rank holds its own blocks plus read-only halo copies of the neighbouring ranks' blocks
(``MeshData.resident_gids``), so a refresh has two kinds of destination cells:

* every ghost cell of every resident block -- mean of the 2^ndim sample points of
  ``Mesh.ghost_sources`` (copy / injection / volume average);
* every interior cell of a halo copy -- copy of the owner's cell.

All sources are INTERIOR cells of the block's OWNER, so one round suffices: each rank packs the
cells other ranks asked for (``jb_gather_cells``), one all-to-all-v moves them (xGMI: every pair
of ranks has its own link), and ``jb_fill_cells`` writes the destinations from local cells and the
received buffer.  The request lists are static for a mesh and are exchanged once, at set-up.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib


class FieldExchange:
    """Plan + buffers for ``refresh(name)`` on one ``MeshData``."""

    def __init__(self, md):
        self.md = md
        mesh = md.mesh
        rank = md.rank
        # (a replicated-mesh run holds every block on every rank: the refresh is local)
        nranks = 1 if getattr(md, "replicated", False) else md.nranks
        owner = np.asarray(getattr(md, "field_owner", mesh.owner), dtype=np.int64)
        self.nranks = nranks
        ns = 2 ** mesh.ndim
        ni, nj = mesh.ntot_dim[0], mesh.ntot_dim[1]
        K, J, I = np.meshgrid(np.arange(mesh.is_[2], mesh.is_[2] + mesh.nx[2]),
                              np.arange(mesh.is_[1], mesh.is_[1] + mesh.nx[1]),
                              np.arange(mesh.is_[0], mesh.is_[0] + mesh.nx[0]), indexing="ij")
        interior_cells = ((K * nj + J) * ni + I).ravel().astype(np.int64)

        dst_blk: List[np.ndarray] = []
        dst_cell: List[np.ndarray] = []
        src_gid: List[np.ndarray] = []
        src_cell: List[np.ndarray] = []
        for lb, g in enumerate(md.resident_gids):
            d, sg, sc = mesh.ghost_sources(int(g))
            dst_blk.append(np.full(len(d), lb, dtype=np.int64))
            dst_cell.append(d)
            src_gid.append(sg)
            src_cell.append(sc)
            if not md.owned_flags[lb]:      # halo copy: its interior mirrors the owner's
                n = len(interior_cells)
                dst_blk.append(np.full(n, lb, dtype=np.int64))
                dst_cell.append(interior_cells)
                src_gid.append(np.full((n, ns), int(g), dtype=np.int64))
                src_cell.append(np.repeat(interior_cells[:, None], ns, axis=1))
        self.nsamples = ns
        dblk = np.concatenate(dst_blk) if dst_blk else np.zeros(0, dtype=np.int64)
        dcell = np.concatenate(dst_cell) if dst_cell else np.zeros(0, dtype=np.int64)
        sgid = np.concatenate(src_gid) if src_gid else np.zeros((0, ns), dtype=np.int64)
        scell = np.concatenate(src_cell) if src_cell else np.zeros((0, ns), dtype=np.int64)
        self.ndst = len(dblk)

        # local sources address the resident copy of the OWNED block; everything else is a request
        src_owner = owner[sgid]
        is_local = src_owner == rank
        sblk = np.where(is_local, md.local_index[sgid], -1).astype(np.int64)
        sidx = scell.copy()
        self.requests: List[np.ndarray] = []     # per source rank: unique (gid * ntot + cell)
        offset = 0
        key = sgid * mesh.ntot + scell
        for r in range(nranks):
            sel = (src_owner == r) & ~is_local
            if r == rank or not sel.any():
                self.requests.append(np.zeros(0, dtype=np.int64))
                continue
            uniq, inv = np.unique(key[sel], return_inverse=True)
            sidx[sel] = offset + inv
            self.requests.append(uniq.astype(np.int64))
            offset += len(uniq)
        self.nremote = offset
        self.recv_counts = np.array([len(q) for q in self.requests], dtype=np.int64)

        # tell every owner what to send us, once
        if nranks > 1:
            served = md.comm.exchange_int64_lists(self.requests)
        else:
            served = [np.zeros(0, dtype=np.int64)]
        self.send_counts = np.array([len(q) for q in served], dtype=np.int64)
        serve = np.concatenate(served) if served else np.zeros(0, dtype=np.int64)
        serve_gid = serve // mesh.ntot
        if len(serve) and not np.all(owner[serve_gid] == rank):
            raise RuntimeError("a rank was asked for cells of a block it does not own")
        dev = md.device
        i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)
        self.serve_blk = i32(md.local_index[serve_gid])
        self.serve_cell = i32(serve % mesh.ntot)
        self.dst_blk, self.dst_cell = i32(dblk), i32(dcell)
        self.src_blk, self.src_cell = i32(sblk.ravel()), i32(sidx.ravel())
        self.send_buf = torch.empty(max(1, len(serve)), dtype=torch.float64, device=dev)
        self.remote = torch.empty(max(1, self.nremote), dtype=torch.float64, device=dev)

    def refresh(self, name: str) -> None:
        md = self.md
        fid = _lib.FIELD_IDS[name]
        md._sync_stream()
        nsend = int(self.send_counts.sum())
        if self.nranks > 1:
            _lib.check(md.lib.jb_gather_cells(md.pkg.ctx, md.handle, fid, nsend,
                                              self.serve_blk.data_ptr(), self.serve_cell.data_ptr(),
                                              self.send_buf.data_ptr()))
            torch.cuda.current_stream(md.device).synchronize()
            md.comm.exchange_values(self.send_buf[:nsend], self.send_counts,
                                    self.remote[:self.nremote], self.recv_counts)
        _lib.check(md.lib.jb_fill_cells(md.pkg.ctx, md.handle, fid, self.ndst, self.nsamples,
                                        self.dst_blk.data_ptr(), self.dst_cell.data_ptr(),
                                        self.src_blk.data_ptr(), self.src_cell.data_ptr(),
                                        self.remote.data_ptr()))

    # test hook: the same plan evaluated with numpy on host copies (no device work)
    def refresh_numpy(self, field: np.ndarray, remote: np.ndarray) -> None:
        flat = field.reshape(field.shape[0], -1)
        sb = self.src_blk.cpu().numpy().reshape(-1, self.nsamples)
        sc = self.src_cell.cpu().numpy().reshape(-1, self.nsamples)
        samples = [np.where(sb[:, q] >= 0, flat[np.maximum(sb[:, q], 0), np.where(sb[:, q] >= 0, sc[:, q], 0)],
                            remote[np.where(sb[:, q] < 0, sc[:, q], 0)] if len(remote) else 0.0)
                   for q in range(self.nsamples)]
        while len(samples) > 1:
            samples = [samples[q] + samples[q + 1] for q in range(0, len(samples), 2)]
        flat[self.dst_blk.cpu().numpy(), self.dst_cell.cpu().numpy()] = samples[0] / self.nsamples
