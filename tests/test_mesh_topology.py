"""An independent check of the SMR topology the history loop runs on (the oracle
and the product both take mesh, leaf map and neighbour levels from jaybenne_amd.mesh, so a wrong
table would be common-mode).  tests/golden/smr_topology.json holds the leaf-block lists of
reference inputs/stepdiff_smr.in (20 blocks, 2 levels) and of BASELINE configs[4]'s 3-level
extension, written by hand from the decks; everything else -- bounds, neighbour levels across the
faces, the finest-level leaf map -- is derived HERE by brute force over that list, without any code
of jaybenne_amd.mesh, and Mesh.from_deck is held to it."""
import json
import os

import numpy as np
import pytest

from helpers import load_deck
from jaybenne_amd.mesh import Mesh

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "smr_topology.json")
CASES = json.load(open(FIXTURE))


def _expected(case):
    blocks = case["blocks"]
    ext = np.array(case["block_extent_level0"])
    gmin = np.array(case["domain_min"])
    nroot = np.array(case["root_blocks"])
    gmax = gmin + nroot * ext
    lo = np.array([gmin + np.array(b[1:]) * ext / 2 ** b[0] for b in blocks])
    hi = np.array([gmin + (np.array(b[1:]) + 1) * ext / 2 ** b[0] for b in blocks])
    level = np.array([b[0] for b in blocks])

    def owner_of(p):            # the one block whose (half-open) box holds the point
        inside = np.all((lo <= p) & (p < hi), axis=1)
        assert inside.sum() == 1, (p, np.nonzero(inside)[0])
        return int(np.nonzero(inside)[0][0])

    # the blocks tile the domain exactly once
    area = np.prod(hi - lo, axis=1).sum()
    assert abs(area - np.prod(gmax - gmin)) < 1e-14
    # neighbour level across each face: probe just outside the face, next to the block's lower corner
    # (so that a finer neighbour is seen at its own level); x outflow: own level, y periodic
    fine = ext / 2 ** level.max()
    nbr = np.empty((len(blocks), 4), dtype=int)
    for b in range(len(blocks)):
        for d in range(2):
            for side in range(2):
                p = lo[b] + 0.25 * fine
                p[d] = lo[b][d] - 0.25 * fine[d] if side == 0 else hi[b][d] + 0.25 * fine[d]
                if d == 0 and (p[0] < gmin[0] or p[0] > gmax[0]):
                    nbr[b, 2 * d + side] = level[b]
                    continue
                if d == 1:
                    p[1] = gmin[1] + (p[1] - gmin[1]) % (gmax[1] - gmin[1])
                nbr[b, 2 * d + side] = level[owner_of(p)]
    nleaf = nroot * 2 ** level.max()
    leaf = np.array([[owner_of(gmin + (np.array([i, j]) + 0.5) * fine) for i in range(nleaf[0])]
                     for j in range(nleaf[1])])
    return lo, hi, level, nbr, nleaf, leaf


@pytest.mark.parametrize("name", ["stepdiff_smr", "three_level"])
def test_smr_topology_matches_the_hand_written_block_list(name):
    case = CASES[name]
    pin = load_deck(case["deck"])
    if case["extra"]:
        pin.load_string(case["extra"])
    mesh = Mesh.from_deck(pin)
    lo, hi, level, nbr, nleaf, leaf = _expected(case)
    assert mesh.nblocks == len(case["blocks"])
    assert np.array_equal(mesh.blk_level, level)
    assert np.array_equal(mesh.blk_lloc[:, :2], np.array([b[1:] for b in case["blocks"]]))
    assert np.array_equal(mesh.blk_xmin[:, :2], lo) and np.array_equal(mesh.blk_xmax[:, :2], hi)
    assert np.array_equal(mesh.blk_nbr_lev[:, :4], nbr)
    assert np.all(mesh.blk_nbr_lev[:, 4:] == mesh.blk_level[:, None])       # inactive dimension
    assert list(mesh.nleaf) == [int(nleaf[0]), int(nleaf[1]), 1]
    assert np.array_equal(mesh.leaf_map[0], leaf)
    # 2:1 balance across faces
    assert np.abs(mesh.blk_nbr_lev[:, :4] - mesh.blk_level[:, None]).max() <= 1


def test_three_level_deck_is_the_bench_workload():
    """The hand-written 3-level list is the mesh bench.py --workload c5 runs."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    mesh = Mesh.from_deck(bench.make_deck(1, 1000, workload="c5"))
    case = CASES["three_level"]
    assert mesh.nblocks == len(case["blocks"]) == 32
    assert np.array_equal(mesh.blk_level, np.array([b[0] for b in case["blocks"]]))
    assert np.array_equal(mesh.blk_lloc[:, :2], np.array([b[1:] for b in case["blocks"]]))


def test_uniform_3d_periodic_mesh_matches_the_listed_neighbour_table():
    """BASELINE configs[1]'s mesh (4 x 4 x 4 blocks of 64^3, periodic in x2 / x3): block order, logical
    locations, the leaf map and the destination block behind every face against
    tests/golden/c2_block_neighbours.json, a table built from the deck's geometry alone
    (tests/golden/make_c2_neighbours.py imports nothing of this package)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    tab = json.load(open(os.path.join(os.path.dirname(FIXTURE), "c2_block_neighbours.json")))
    mesh = Mesh.from_deck(bench.make_deck(1, 1000, workload="c2"))
    assert mesh.nblocks == 64 and mesh.ndim == 3 and list(mesh.nx) == [64, 64, 64]
    ext = np.array(tab["block_extent"])
    gmin = np.array(tab["domain_min"])
    for b, ent in enumerate(tab["blocks"]):
        lloc = np.array(ent["lloc"])
        assert np.array_equal(mesh.blk_lloc[b], lloc) and mesh.blk_level[b] == 0
        assert np.array_equal(mesh.blk_xmin[b], gmin + lloc * ext)
        assert np.array_equal(mesh.blk_xmax[b], gmin + (lloc + 1) * ext)
        assert mesh.leaf_map[lloc[2], lloc[1], lloc[0]] == b
        # where a photon that has just crossed a face is handed to: a point a quarter of a cell beyond it
        # (wrapped where the boundary is periodic), through the package's own destination look-up
        ctr = gmin + (lloc + 0.5) * ext
        for f, want in enumerate(ent["faces"]):
            d, up = f >> 1, f & 1
            p = ctr.copy()
            p[d] = (gmin[d] + (lloc[d] + up) * ext[d]) + (0.25 / 256) * (1 if up else -1)
            if p[d] < gmin[d] or p[d] > gmin[d] + 4 * ext[d]:
                if want < 0:
                    assert d == 0
                    continue
                p[d] += 4 * ext[d] * (1 if p[d] < gmin[d] else -1)
            assert want >= 0 and mesh.find_block(p[None, :])[0] == want, (b, f)
    # every neighbour relation is mutual, and Mesh.neighbours (the halo copies of a rank) agrees with the
    # table's faces plus edges and corners: 26 distinct blocks around an interior block
    assert set(int(g) for g in mesh.neighbours([21])) >= {g for g in tab["blocks"][21]["faces"] if g >= 0}
    assert len(mesh.neighbours([21])) == 26


@pytest.mark.parametrize("deck,overrides,ndim,mesh_nx,block_nx,gmin,gmax,box,periodic", [
    # the 3-D SMR deck of the parity tests (tests/test_gpu_parity.py: SMR3D): 4 x 2 x 2 root blocks of 8^3, the
    # middle of the domain at level 1 -> 72 blocks
    ("stepdiff_smr_ddmc", {"parthenon/mesh/nx1": 32, "parthenon/mesh/nx2": 16, "parthenon/mesh/nx3": 16,
                           "parthenon/meshblock/nx1": 8, "parthenon/meshblock/nx2": 8, "parthenon/meshblock/nx3": 8},
     3, (32, 16, 16), (8, 8, 8), (-0.5, -0.25, -0.25), (0.5, 0.25, 0.25),
     ((-0.25, 0.25), (-0.25, 0.25), (-0.25, 0.25)), (False, True, True)),
    # the 2-D SMR deck at the parity tests' size
    ("stepdiff_smr_ddmc", {"parthenon/mesh/nx1": 64, "parthenon/mesh/nx2": 32,
                           "parthenon/meshblock/nx1": 16, "parthenon/meshblock/nx2": 16},
     2, (64, 32, 1), (16, 16, 1), (-0.5, -0.25, -0.25), (0.5, 0.25, 0.25),
     ((-0.25, 0.25), (-0.25, 0.25), (-0.25, 0.25)), (False, True, True)),
    # 1-D in 32 blocks of 4 cells (tests/test_gpu_multirank.py)
    ("stepdiff_ddmc", {"parthenon/mesh/nx1": 128, "parthenon/meshblock/nx1": 4},
     1, (128, 1, 1), (4, 1, 1), (-0.5, -0.5, -0.5), (0.5, 0.5, 0.5), None, (False, True, True)),
])
def test_mesh_against_the_independent_builder_of_the_oracle(deck, overrides, ndim, mesh_nx, block_nx, gmin, gmax,
                                                            box, periodic):
    """oracle/meshcheck.py (no code of jaybenne_amd; geometry typed in here from the decks) against
    Mesh.from_deck on the decks the hand-written fixtures above do not cover: the same set of leaf blocks
    (level, logical location), the same bounds, the same neighbour level behind every face -- the tables both
    the oracle and the product are handed."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import meshcheck
    mesh = Mesh.from_deck(load_deck(deck, overrides))
    assert mesh.ndim == ndim
    blocks = meshcheck.leaf_blocks(ndim, mesh_nx, block_nx, gmin, gmax, box)
    nbr = meshcheck.neighbour_levels(ndim, blocks, gmin, gmax, periodic)
    assert mesh.nblocks == len(blocks)
    mine = {(int(mesh.blk_level[b]), tuple(int(v) for v in mesh.blk_lloc[b])): b for b in range(mesh.nblocks)}
    assert len(mine) == mesh.nblocks
    for lev, loc, lo, hi in blocks:
        assert (lev, loc) in mine, (lev, loc)
        b = mine[(lev, loc)]
        assert np.array_equal(mesh.blk_xmin[b, :ndim], np.array(lo[:ndim])), (lev, loc)
        assert np.array_equal(mesh.blk_xmax[b, :ndim], np.array(hi[:ndim])), (lev, loc)
        assert [int(v) for v in mesh.blk_nbr_lev[b]] == nbr[(lev, loc)], (lev, loc)
    if box is not None:
        assert len(set(b[0] for b in blocks)) == 2
