"""Block-structured mesh with static refinement: the part of Parthenon's ``Mesh`` the history
loop depends on (block list, per-block geometry, destination-block lookup, neighbour levels,
ghost-zone fill, block -> rank partition).

Parthenon itself is an empty submodule in the reference tree, so its semantics are restated from
the call sites:

* interior index range ``ng .. ng+nx-1`` in active dimensions, a single index ``0`` otherwise
  (reference transport.cpp:57-59, sourcing.cpp:59-66);
* ``<parthenon/mesh>``, ``<parthenon/meshblock>``, ``<parthenon/static_refinementN>`` and
  ``<parthenon/swarm>`` deck blocks (reference inputs/stepdiff_smr.in:16-59);
* a particle that leaves its block is handed to the leaf block that contains it
  (``GetNeighborBlockIndex`` + ``Swarm::Send``, reference transport.cpp:149-155,
  jaybenne.cpp:26-61).  Here that lookup is a table over blocks of the finest level
  (``leaf_map``) instead of Parthenon's per-block 4x4x4 neighbour table: same destination,
  one gather.
"""
from __future__ import annotations

import itertools
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

BC_PERIODIC, BC_REFLECT, BC_OUTFLOW = 0, 1, 2
_SWARM_BC = {"periodic": BC_PERIODIC, "jaybenne_reflecting": BC_REFLECT, "outflow": BC_OUTFLOW}


@dataclass
class Refinement:
    level: int
    lo: Tuple[float, float, float]
    hi: Tuple[float, float, float]


def _morton(l: Sequence[int], bits: int) -> int:
    key = 0
    for b in range(bits):
        for d in range(3):
            key |= ((int(l[d]) >> b) & 1) << (3 * b + d)
    return key


class Mesh:
    """Leaf blocks in Z-order with everything the kernels need as flat numpy arrays."""

    def __init__(self, ndim: int, mesh_nx: Sequence[int], block_nx: Sequence[int],
                 gmin: Sequence[float], gmax: Sequence[float], ng: int = 2,
                 mesh_bc: Sequence[int] = (BC_OUTFLOW, BC_OUTFLOW, BC_PERIODIC, BC_PERIODIC,
                                           BC_PERIODIC, BC_PERIODIC),
                 swarm_bc: Sequence[int] = (BC_REFLECT, BC_REFLECT, BC_PERIODIC, BC_PERIODIC,
                                            BC_PERIODIC, BC_PERIODIC),
                 refinements: Sequence[Refinement] = ()):
        self.ndim = int(ndim)
        self.mesh_nx = [int(v) for v in mesh_nx]
        self.nx = [int(v) for v in block_nx]
        for d in range(3):
            if d >= self.ndim and (self.mesh_nx[d] != 1 or self.nx[d] != 1):
                raise ValueError("inactive dimensions must have nx = 1")
            if self.mesh_nx[d] % self.nx[d] != 0:
                raise ValueError("mesh size must be a multiple of the block size")
        if ng < 1:
            raise ValueError("at least one ghost layer is needed (face fields share the cell layout)")
        self.ng = int(ng)
        self.gmin = np.asarray(gmin, dtype=np.float64).copy()
        self.gmax = np.asarray(gmax, dtype=np.float64).copy()
        self.mesh_bc = [int(v) for v in mesh_bc]
        self.swarm_bc = [int(v) for v in swarm_bc]
        self.nroot = [self.mesh_nx[d] // self.nx[d] for d in range(3)]
        self.refinements = list(refinements)
        self._build_tree()
        self._build_arrays()

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_deck(cls, pin) -> "Mesh":
        mb = "parthenon/mesh"
        mesh_nx = [pin.GetInteger(mb, f"nx{d}") for d in (1, 2, 3)]
        ndim = 1 + int(mesh_nx[1] > 1) + int(mesh_nx[2] > 1)
        if mesh_nx[2] > 1 and mesh_nx[1] == 1:
            raise ValueError("nx3 > 1 requires nx2 > 1")
        gmin = [pin.GetReal(mb, f"x{d}min") for d in (1, 2, 3)]
        gmax = [pin.GetReal(mb, f"x{d}max") for d in (1, 2, 3)]
        block_nx = [pin.GetOrAddInteger("parthenon/meshblock", f"nx{d}", mesh_nx[d - 1])
                    for d in (1, 2, 3)]
        ng = pin.GetOrAddInteger(mb, "nghost", 2)

        def bcs(block, default):
            out = []
            for d in (1, 2, 3):
                for side in ("i", "o"):
                    name = pin.GetOrAddString(block, f"{side}x{d}_bc", default)
                    if name not in _SWARM_BC:
                        raise ValueError(f"unsupported boundary condition '{name}' in <{block}>")
                    out.append(_SWARM_BC[name])
            return out

        mesh_bc = bcs(mb, "periodic")
        # a reflecting *mesh* boundary is not used by any deck; treat non-periodic as outflow
        mesh_bc = [BC_PERIODIC if b == BC_PERIODIC else BC_OUTFLOW for b in mesh_bc]
        swarm_bc = bcs("parthenon/swarm", "periodic")
        refs = []
        if pin.GetOrAddString(mb, "refinement", "none") == "static":
            n = 1
            while pin.DoesBlockExist(f"parthenon/static_refinement{n}"):
                rb = f"parthenon/static_refinement{n}"
                lo = [pin.GetOrAddReal(rb, f"x{d}min", gmin[d - 1]) for d in (1, 2, 3)]
                hi = [pin.GetOrAddReal(rb, f"x{d}max", gmax[d - 1]) for d in (1, 2, 3)]
                refs.append(Refinement(pin.GetInteger(rb, "level"), tuple(lo), tuple(hi)))
                n += 1
        return cls(ndim, mesh_nx, block_nx, gmin, gmax, ng, mesh_bc, swarm_bc, refs)

    def _block_bounds(self, level: int, lx: Sequence[int]):
        lo = np.empty(3)
        hi = np.empty(3)
        for d in range(3):
            nb = self.nroot[d] * (1 << level if d < self.ndim else 1)
            ext = self.gmax[d] - self.gmin[d]
            lo[d] = self.gmin[d] + ext * (lx[d] / nb)
            hi[d] = self.gmin[d] + ext * ((lx[d] + 1) / nb)
        return lo, hi

    def _children(self, level: int, lx):
        rng = [(0, 1) if d < self.ndim else (0,) for d in range(3)]
        return [(level + 1, tuple(2 * lx[d] + o[d] if d < self.ndim else lx[d] for d in range(3)))
                for o in itertools.product(*rng)]

    def _build_tree(self) -> None:
        leaves = {(0, (i, j, k)) for k in range(self.nroot[2]) for j in range(self.nroot[1])
                  for i in range(self.nroot[0])}
        max_level = max([r.level for r in self.refinements], default=0)
        # refine every leaf that overlaps (open overlap) a region of higher level
        for target in range(1, max_level + 1):
            for (lev, lx) in sorted(leaves):
                if lev != target - 1:
                    continue
                lo, hi = self._block_bounds(lev, lx)
                for r in self.refinements:
                    if r.level < target:
                        continue
                    if all(lo[d] < r.hi[d] and hi[d] > r.lo[d] for d in range(self.ndim)):
                        leaves.remove((lev, lx))
                        leaves.update(self._children(lev, lx))
                        break
        # 2:1 balance across faces, edges and corners
        changed = True
        while changed:
            changed = False
            max_level = max(l for l, _ in leaves)
            lmap = self._leaf_map_from(leaves, max_level)
            for (lev, lx) in sorted(leaves):
                if lev < 2:
                    continue
                scale = 1 << (max_level - lev)
                offs = [(-1, 0, 1) if d < self.ndim else (0,) for d in range(3)]
                for o in itertools.product(*offs):
                    # same-level logical location of the neighbour region, mapped to the
                    # finest-level block grid of the leaf map
                    q = []
                    ok = True
                    for d in range(3):
                        n = lmap["shape"][d]
                        c = (lx[d] + o[d]) * scale if d < self.ndim else 0
                        if c < 0 or c >= n:
                            if self.mesh_bc[2 * d] == BC_PERIODIC:
                                c %= n
                            else:
                                ok = False
                        q.append(c)
                    if not ok:
                        continue
                    nb = lmap["leaf"][q[2]][q[1]][q[0]]
                    if nb[0] < lev - 1 and nb in leaves:
                        leaves.remove(nb)
                        leaves.update(self._children(*nb))
                        changed = True
                if changed:
                    break
        self.max_level = max(l for l, _ in leaves)
        bits = max(1, int(np.ceil(np.log2(max(self.nroot) * (1 << self.max_level) + 1))))
        self.leaves: List[Tuple[int, Tuple[int, int, int]]] = sorted(
            leaves,
            key=lambda b: _morton([b[1][d] << (self.max_level - b[0]) if d < self.ndim else 0
                                   for d in range(3)], bits))

    def _leaf_map_from(self, leaves, max_level):
        shape = [self.nroot[d] * ((1 << max_level) if d < self.ndim else 1) for d in range(3)]
        leaf = [[[None] * shape[0] for _ in range(shape[1])] for _ in range(shape[2])]
        for (lev, lx) in leaves:
            s = 1 << (max_level - lev)
            r = [range(lx[d] * s, (lx[d] + 1) * s) if d < self.ndim else range(1) for d in range(3)]
            for k in r[2]:
                for j in r[1]:
                    for i in r[0]:
                        leaf[k][j][i] = (lev, lx)
        return {"shape": shape, "leaf": leaf}

    def _build_arrays(self) -> None:
        nb = len(self.leaves)
        self.nblocks = nb
        self.blk_level = np.array([l for l, _ in self.leaves], dtype=np.int32)
        self.blk_lloc = np.array([lx for _, lx in self.leaves], dtype=np.int32)
        self.blk_xmin = np.empty((nb, 3))
        self.blk_xmax = np.empty((nb, 3))
        for b, (lev, lx) in enumerate(self.leaves):
            self.blk_xmin[b], self.blk_xmax[b] = self._block_bounds(lev, lx)
        self.blk_dx = (self.blk_xmax - self.blk_xmin) / np.asarray(self.nx, dtype=np.float64)
        self.nleaf = [self.nroot[d] * ((1 << self.max_level) if d < self.ndim else 1)
                      for d in range(3)]
        index = {blk: b for b, blk in enumerate(self.leaves)}
        lm = self._leaf_map_from(set(self.leaves), self.max_level)["leaf"]
        self.leaf_map = np.array([[[index[lm[k][j][i]] for i in range(self.nleaf[0])]
                                   for j in range(self.nleaf[1])] for k in range(self.nleaf[2])],
                                 dtype=np.int32)
        # index space
        self.is_ = [self.ng if d < self.ndim else 0 for d in range(3)]
        self.ntot_dim = [self.nx[d] + 2 * self.is_[d] for d in range(3)]   # (ni, nj, nk)
        self.ntot = int(np.prod(self.ntot_dim))
        self.ncell = int(np.prod(self.nx))
        self.field_shape = (nb, self.ntot_dim[2], self.ntot_dim[1], self.ntot_dim[0])
        # neighbour levels across the six faces (own level at physical boundaries)
        self.blk_nbr_lev = np.empty((nb, 6), dtype=np.int32)
        fine = (self.gmax - self.gmin) / np.asarray(self.nleaf, dtype=np.float64)
        for b in range(nb):
            ctr = 0.5 * (self.blk_xmin[b] + self.blk_xmax[b])
            for d in range(3):
                for side in (0, 1):
                    if d >= self.ndim:
                        self.blk_nbr_lev[b, 2 * d + side] = self.blk_level[b]
                        continue
                    p = ctr + 0.25 * fine * np.array([dd < self.ndim for dd in range(3)])
                    p[d] = (self.blk_xmin[b, d] - 0.25 * fine[d]) if side == 0 else \
                        (self.blk_xmax[b, d] + 0.25 * fine[d])
                    if p[d] < self.gmin[d] or p[d] > self.gmax[d]:
                        if self.mesh_bc[2 * d + side] != BC_PERIODIC:
                            self.blk_nbr_lev[b, 2 * d + side] = self.blk_level[b]
                            continue
                        p[d] += (self.gmax[d] - self.gmin[d]) * (1 if side == 0 else -1)
                    self.blk_nbr_lev[b, 2 * d + side] = self.blk_level[self.find_block(p[None, :])[0]]
        self.owner = np.zeros(nb, dtype=np.int32)

    # ------------------------------------------------------------------ geometry helpers
    def find_block(self, pts: np.ndarray) -> np.ndarray:
        """Leaf block containing each point (points must lie inside the domain)."""
        q = []
        for d in range(3):
            ln = (self.gmax[d] - self.gmin[d]) / self.nleaf[d]
            q.append(np.clip(np.floor((pts[:, d] - self.gmin[d]) / ln).astype(np.int64), 0,
                             self.nleaf[d] - 1))
        return self.leaf_map[q[2], q[1], q[0]]

    def cell_centers(self, b: int, d: int) -> np.ndarray:
        """Xc(idx) for every index (ghosts included) of dimension d of block b."""
        dx = self.blk_dx[b, d]
        x0 = self.blk_xmin[b, d] - self.is_[d] * dx
        return x0 + (np.arange(self.ntot_dim[d]) + 0.5) * dx

    def interior(self) -> Tuple[slice, slice, slice, slice]:
        s = [slice(self.is_[d], self.is_[d] + self.nx[d]) for d in range(3)]
        return (slice(None), s[2], s[1], s[0])

    def cell_volume(self, b: int) -> float:
        return float(self.blk_dx[b, 0] * self.blk_dx[b, 1] * self.blk_dx[b, 2])

    def new_field(self, fill: float = 0.0) -> np.ndarray:
        return np.full(self.field_shape, fill, dtype=np.float64)

    # ------------------------------------------------------------------ ghost zones
    def ghost_sources(self, b: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """For every ghost cell of block ``b``: where its value comes from.  Returns
        ``(dst, src_gid, src_cell)``: ``dst[n]`` is the ghost cell (flat index into the block's
        ``[nk][nj][ni]`` array), ``src_gid[n, s]`` / ``src_cell[n, s]`` the interior cell of the
        leaf block that owns sample point ``s`` (2^ndim points at +-dx/4 around the ghost's
        centre): all equal for a same-level or coarser neighbour (copy / injection), the 2^ndim
        children for a finer one (volume average).  Periodic wrap or nearest-interior copy
        (outflow) at the domain boundary.  This is the static part of Parthenon's
        ``AddBoundaryExchangeTasks`` on ``FillGhost`` fields (reference mcblock_driver.cpp:68,
        mcblock.cpp:68-71); Parthenon prolongates coarse data with limited linear interpolation,
        injection differs from it only where the field varies inside a coarse cell's
        neighbourhood."""
        offs = list(itertools.product(*[(-0.25, 0.25) if d < self.ndim else (0.0,)
                                        for d in range(3)]))
        xs = [self.cell_centers(b, d) for d in range(3)]
        K, J, I = np.meshgrid(np.arange(self.ntot_dim[2]), np.arange(self.ntot_dim[1]),
                              np.arange(self.ntot_dim[0]), indexing="ij")
        ghost = np.zeros(K.shape, dtype=bool)
        for d, idx in ((0, I), (1, J), (2, K)):
            if d < self.ndim:
                ghost |= (idx < self.is_[d]) | (idx >= self.is_[d] + self.nx[d])
        kk, jj, ii = K[ghost], J[ghost], I[ghost]
        ni, nj = self.ntot_dim[0], self.ntot_dim[1]
        dst = (kk * nj + jj) * ni + ii
        src_gid = np.empty((len(ii), len(offs)), dtype=np.int64)
        src_cell = np.empty((len(ii), len(offs)), dtype=np.int64)
        if len(ii) == 0:
            return dst.astype(np.int64), src_gid, src_cell
        base = np.stack([xs[0][ii], xs[1][jj], xs[2][kk]], axis=1)
        for q, o in enumerate(offs):
            pt = base + np.asarray(o) * self.blk_dx[b]
            for d in range(self.ndim):
                ext = self.gmax[d] - self.gmin[d]
                if self.mesh_bc[2 * d] == BC_PERIODIC:
                    pt[:, d] = np.where(pt[:, d] < self.gmin[d], pt[:, d] + ext, pt[:, d])
                else:
                    pt[:, d] = np.maximum(pt[:, d], self.gmin[d] + 0.25 * self.blk_dx[b, d])
                if self.mesh_bc[2 * d + 1] == BC_PERIODIC:
                    pt[:, d] = np.where(pt[:, d] > self.gmax[d], pt[:, d] - ext, pt[:, d])
                else:
                    pt[:, d] = np.minimum(pt[:, d], self.gmax[d] - 0.25 * self.blk_dx[b, d])
            nbk = self.find_block(pt)
            c = []
            for d in range(3):
                if d < self.ndim:
                    cd = np.floor((pt[:, d] - self.blk_xmin[nbk, d]) / self.blk_dx[nbk, d])
                    cd = np.clip(cd.astype(np.int64), 0, self.nx[d] - 1) + self.is_[d]
                else:
                    cd = np.zeros(len(ii), dtype=np.int64)
                c.append(cd)
            src_gid[:, q] = nbk
            src_cell[:, q] = (c[2] * nj + c[1]) * ni + c[0]
        return dst.astype(np.int64), src_gid, src_cell

    def fill_ghosts(self, f: np.ndarray) -> None:
        """Fill the ghost cells of a cell-centred field (whole mesh, host arrays) from the
        source map of ``ghost_sources``; the samples are summed pairwise, which is exact when
        they are equal (same-level or coarser neighbour)."""
        if not f.flags.c_contiguous:
            raise ValueError("fill_ghosts needs a C-contiguous field")
        src = f.reshape(self.nblocks, -1).copy()
        out = f.reshape(self.nblocks, -1)
        for b in range(self.nblocks):
            dst, gid, cell = self.ghost_sources(b)
            if len(dst) == 0:
                continue
            samples = [src[gid[:, q], cell[:, q]] for q in range(gid.shape[1])]
            while len(samples) > 1:
                samples = [samples[q] + samples[q + 1] for q in range(0, len(samples), 2)]
            out[b, dst] = samples[0] / gid.shape[1]

    # ------------------------------------------------------------------ halo
    def neighbours(self, gids: Sequence[int], rings: int = 1) -> np.ndarray:
        """Leaf blocks that touch (face, edge or corner; through periodic boundaries too) any of
        the blocks `gids`, excluding those blocks; `rings` repeats the growth."""
        have = set(int(g) for g in gids)
        frontier = list(have)
        fine = (self.gmax - self.gmin) / np.asarray(self.nleaf, dtype=np.float64)
        for _ in range(rings):
            new = set()
            for b in frontier:
                axes = []
                for d in range(3):
                    if d >= self.ndim:
                        axes.append(np.array([0.5 * (self.gmin[d] + self.gmax[d])]))
                        continue
                    lo = self.blk_xmin[b, d] - 0.5 * fine[d]
                    n = int(round((self.blk_xmax[b, d] - self.blk_xmin[b, d]) / fine[d])) + 2
                    c = lo + fine[d] * np.arange(n)
                    ext = self.gmax[d] - self.gmin[d]
                    if self.mesh_bc[2 * d] == BC_PERIODIC:
                        c = np.where(c < self.gmin[d], c + ext, c)
                    if self.mesh_bc[2 * d + 1] == BC_PERIODIC:
                        c = np.where(c > self.gmax[d], c - ext, c)
                    axes.append(c[(c > self.gmin[d]) & (c < self.gmax[d])])
                X, Y, Z = np.meshgrid(*axes, indexing="ij")
                pts = np.column_stack([X.ravel(), Y.ravel(), Z.ravel()])
                new.update(int(g) for g in np.unique(self.find_block(pts)))
            new -= have
            have |= new
            frontier = list(new)
        return np.array(sorted(have - set(int(g) for g in gids)), dtype=np.int32)

    # ------------------------------------------------------------------ partition
    def sibling_groups(self) -> np.ndarray:
        """``group[b]``: the same number for the blocks that are children of one parent and follow each
        other in the Z-ordered list (a refined block's 2^ndim children), a number of its own for every
        other block."""
        group = np.empty(self.nblocks, dtype=np.int64)
        g = -1
        prev = None
        for b, (lev, lx) in enumerate(self.leaves):
            key = (lev, tuple(lx[d] >> 1 if d < self.ndim else 0 for d in range(3))) if lev > 0 else None
            if key is None or key != prev:
                g += 1
            group[b] = g
            prev = key
        return group

    def partition(self, nranks: int, cost: Optional[Sequence[float]] = None,
                  sibling_slack: float = 0.10) -> np.ndarray:
        """Contiguous runs of the Z-ordered block list per rank.  Returns ``owner[b]``.

        ``cost=None``: unit cost per block, runs of equal length (Parthenon's default load balance:
        the reference inherits it, jaybenne.cpp:92-95).  With a cost per block -- the tracking work
        ``mcblock.block_costs`` estimates: photons sourced there x events per history -- the runs are
        the contiguous split with the SMALLEST LARGEST load (a cycle lasts as long as its slowest
        rank), found exactly by dynamic programming over the prefix sums; then every boundary that
        separates the children of one parent moves to the nearer end of that family if the largest load
        stays within ``1 + sibling_slack`` of the optimum (children of one parent exchange the most
        photons).  ``include/jaybenne_amd.hpp: PartitionBlocks`` is the same algorithm for the C++
        hosts, operation for operation.  (Cost: nranks x nblocks^2 / 2 comparisons -- a quarter of a second for 4096
        blocks on 8 ranks; done once per mesh.)"""
        nb = self.nblocks
        if nranks > nb:
            raise ValueError(f"cannot spread {nb} blocks over {nranks} ranks")
        if cost is None:
            bounds = [(r * nb) // nranks for r in range(nranks + 1)]
        else:
            bounds = partition_bounds(np.asarray(cost, dtype=np.float64), nranks, self.sibling_groups(),
                                      sibling_slack)
        owner = np.empty(nb, dtype=np.int32)
        for r in range(nranks):
            owner[bounds[r]:bounds[r + 1]] = r
        self.owner = owner
        return owner


def partition_bounds(cost: np.ndarray, nranks: int, group: np.ndarray, sibling_slack: float = 0.10):
    """``bounds[r] .. bounds[r + 1]`` = the blocks of rank r: see ``Mesh.partition``."""
    nb = len(cost)
    if np.any(cost <= 0.0) or not np.all(np.isfinite(cost)):
        raise ValueError("block costs must be positive and finite")
    S = np.concatenate(([0.0], np.cumsum(cost)))          # sequential sums (the C++ host forms the same)
    # best[r][i]: smallest largest load of blocks [0, i) in r + 1 runs; cut[r][i]: where the last run starts
    best = np.full((nranks, nb + 1), np.inf)
    cut = np.zeros((nranks, nb + 1), dtype=np.int64)
    best[0, 1:] = S[1:]
    for r in range(1, nranks):
        for i in range(r + 1, nb + 1):
            j = np.arange(r, i)
            v = np.maximum(best[r - 1, r:i], S[i] - S[r:i])
            k = int(np.argmin(v))                          # (the first of equal minima)
            best[r, i] = v[k]
            cut[r, i] = j[k]
    bounds = [0] * (nranks + 1)
    bounds[nranks] = nb
    for r in range(nranks - 1, 0, -1):
        bounds[r] = int(cut[r, bounds[r + 1]])
    optimum = float(best[nranks - 1, nb])

    def largest(bd):
        return max(S[bd[r + 1]] - S[bd[r]] for r in range(nranks))

    for r in range(1, nranks):
        q = bounds[r]
        if group[q - 1] != group[q]:
            continue                                       # not inside a family
        lo = q
        while lo > 0 and group[lo - 1] == group[q]:
            lo -= 1
        hi = q
        while hi < nb and group[hi] == group[q]:
            hi += 1
        tries = []
        for cand in (lo, hi):
            if bounds[r - 1] < cand < bounds[r + 1]:
                bd = list(bounds)
                bd[r] = cand
                tries.append((largest(bd), abs(cand - q), cand))
        if tries:
            load, _, cand = min(tries)
            if load <= (1.0 + sibling_slack) * optimum:
                bounds[r] = cand
    return bounds
