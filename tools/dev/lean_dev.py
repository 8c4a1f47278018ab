"""Largest deviations of the lean arithmetic variant from the exact one (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import load_deck
from jaybenne_amd import mcblock
ov = {"parthenon/mesh/nx1": 32, "parthenon/mesh/nx2": 32, "parthenon/mesh/nx3": 32,
      "parthenon/meshblock/nx1": 16, "parthenon/meshblock/nx2": 16, "parthenon/meshblock/nx3": 16,
      "jaybenne/num_particles": int(sys.argv[1]) if len(sys.argv) > 1 else 1000000}
out = {}
for mode in ("lean", "exact"):
    drv = mcblock.McblockDriver(load_deck("stepdiff", ov), device=torch.device("cuda", 0))
    drv.pkg.set_arithmetic(mode)
    for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 1):
        drv.Step()
    out[mode] = (drv.md.get_swarm(), drv.md.n, drv.md.events, drv.md.get_field("tally")[drv.mesh.interior()])
    del drv
(g, n, ev, tl), (h, m, ev2, te) = out["lean"], out["exact"]
print("particles", n, m, "events", ev, ev2)
for k in ("ip", "jp", "kp", "blk", "rng", "status"):
    print(k, "differing:", int(np.sum(g[k] != h[k][:n])))
for k, sc in (("x", 1.0), ("y", 1.0), ("z", 1.0), ("vx", 2.99792458e10), ("t", 3.335641e-11)):
    d = np.abs(g[k] - h[k][:n]) / sc
    print(k, "max dev / scale %.3e" % d.max(), "identical: %.4f" % np.mean(g[k] == h[k][:n]))
print("tally max rel dev %.3e" % (np.abs(tl - te).max() / np.abs(te).max()))
