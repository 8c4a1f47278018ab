"""An independent leaf-block builder for uniform and one-level statically refined Cartesian meshes -- TEST
INFRASTRUCTURE (oracle/README.md).  It imports nothing of jaybenne_amd: oracle/harness.py takes mesh, leaf
map and neighbour levels from the product's host code (jaybenne_amd.mesh), so a wrong table there would be
common to oracle and product; tests/test_mesh_topology.py holds Mesh.from_deck to this brute-force
restatement on the decks the hand-written fixtures do not cover (3-D SMR, 1-D in many blocks).

What it restates (reference inputs/*.in): the root grid is mesh_nx / block_nx blocks per
active axis; a `<parthenon/static_refinement*>` box at level 1 replaces every root block it overlaps (open
intervals: touching at a face is no overlap) by its 2^ndim children; a block's neighbour level across a face
is the level of the leaf that holds a point just beyond that face, next to the block's lower corner (so that
a finer neighbour is seen at its own level), through a periodic boundary where the axis is periodic and the
block's own level at any other domain boundary (jaybenne.cpp:346-351)."""
import itertools


def leaf_blocks(ndim, mesh_nx, block_nx, gmin, gmax, box=None):
    """-> list of (level, (lx, ly, lz), lo[3], hi[3]); box = ((x1min, x1max), (x2min, x2max), (x3min, x3max))."""
    nroot = [mesh_nx[d] // block_nx[d] if d < ndim else 1 for d in range(3)]
    ext = [(gmax[d] - gmin[d]) / nroot[d] for d in range(3)]
    out = []
    for loc in itertools.product(*(range(n) for n in reversed(nroot))):
        loc = loc[::-1]
        lo = [gmin[d] + loc[d] * ext[d] for d in range(3)]
        hi = [gmin[d] + (loc[d] + 1) * ext[d] for d in range(3)]
        refine = box is not None and all(lo[d] < box[d][1] and hi[d] > box[d][0] for d in range(ndim))
        if not refine:
            out.append((0, tuple(loc), lo, hi))
            continue
        for child in itertools.product(*(range(2) if d < ndim else range(1) for d in range(3))):
            clo = [lo[d] + child[d] * ext[d] / 2 if d < ndim else lo[d] for d in range(3)]
            chi = [clo[d] + ext[d] / 2 if d < ndim else hi[d] for d in range(3)]
            out.append((1, tuple(2 * loc[d] + child[d] if d < ndim else 0 for d in range(3)), clo, chi))
    return out


def neighbour_levels(ndim, blocks, gmin, gmax, periodic):
    """-> {(level, lloc): [level behind face x-, x+, y-, y+, z-, z+]} (inactive axes: the block's own level)."""
    finest = max(b[0] for b in blocks)

    def owner(p):
        hits = [b for b in blocks if all(b[2][d] <= p[d] < b[3][d] for d in range(ndim))]
        assert len(hits) == 1, (p, hits)
        return hits[0]

    out = {}
    for lev, loc, lo, hi in blocks:
        width = [(hi[d] - lo[d]) / 2 ** (finest - lev + 2) for d in range(3)]   # a quarter of the finest cell scale
        row = []
        for d in range(3):
            for side in range(2):
                if d >= ndim:
                    row.append(lev)
                    continue
                p = [lo[e] + width[e] for e in range(3)]
                p[d] = lo[d] - width[d] if side == 0 else hi[d] + width[d]
                if p[d] < gmin[d] or p[d] >= gmax[d]:
                    if not periodic[d]:
                        row.append(lev)
                        continue
                    p[d] = gmin[d] + (p[d] - gmin[d]) % (gmax[d] - gmin[d])
                row.append(owner(p)[0])
        out[(lev, loc)] = row
    return out
