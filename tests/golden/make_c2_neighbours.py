"""Writes tests/golden/c2_block_neighbours.json: the 64 meshblocks of BASELINE configs[1] (stepdiff on a
256^3 mesh in 4 x 4 x 4 blocks of 64^3) in Z-order with their six face neighbours, from the deck's
geometry alone (reference inputs/stepdiff.in:30-44: x1 outflow mesh / reflecting swarm boundary, x2 and
x3 periodic).  Deliberately self-contained: no import of jaybenne_amd, nothing but the two rules
    id(i, j, k)  = bits i0 j0 k0 i1 j1 k1 interleaved, i lowest (Morton order of the root grid),
    neighbour    = (i +- 1, j, k) without wrap (x walls: -1), (i, j +- 1 mod 4, k), (i, j, k +- 1 mod 4).
tests/test_mesh_topology.py holds Mesh.from_deck (block order, logical locations, leaf map, the
destination of a point just across every face) to the table."""
import json
import os


def morton(i, j, k):
    key = 0
    for b in range(2):
        key |= ((i >> b) & 1) << (3 * b) | ((j >> b) & 1) << (3 * b + 1) | ((k >> b) & 1) << (3 * b + 2)
    return key


blocks = [None] * 64
for k in range(4):
    for j in range(4):
        for i in range(4):
            nb = [morton(i - 1, j, k) if i > 0 else -1, morton(i + 1, j, k) if i < 3 else -1,
                  morton(i, (j - 1) % 4, k), morton(i, (j + 1) % 4, k),
                  morton(i, j, (k - 1) % 4), morton(i, j, (k + 1) % 4)]
            blocks[morton(i, j, k)] = {"lloc": [i, j, k], "faces": nb}
doc = {"_comment": __doc__.split("\n"), "root_blocks": [4, 4, 4], "domain_min": [-0.5, -0.5, -0.5],
       "block_extent": [0.25, 0.25, 0.25], "face_order": ["x-", "x+", "y-", "y+", "z-", "z+"],
       "blocks": blocks}
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "c2_block_neighbours.json"), "w") as fh:
    json.dump(doc, fh, indent=1)
