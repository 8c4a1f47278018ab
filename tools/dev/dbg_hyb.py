import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from helpers import load_deck, make_oracle, run_oracle_cycles
from jaybenne_amd import mcblock
from oracle import orc
ov = {"parthenon/mesh/nx1": 120, "parthenon/mesh/nx2": 60, "parthenon/meshblock/nx1": 30, "parthenon/meshblock/nx2": 30, "jaybenne/num_particles": 30000}
res = {}
for mode in ("exact", "lean"):
    drv = mcblock.McblockDriver(load_deck("stepdiff_smr_hybrid", ov), device=torch.device("cuda", 0))
    drv.pkg.set_arithmetic(mode)
    drv.Step()
    res[mode] = (drv.md.get_swarm(), drv.md.events, drv.md.lib.jb_last_transport_variant(drv.md.handle).decode())
    print(mode, res[mode][1], res[mode][2])
g, h = res["lean"][0], res["exact"][0]
diff = g["rng"] != h["rng"]
print("photons with a different stream state:", int(diff.sum()), "of", len(diff))
idx = np.nonzero(diff)[0][:10]
for i in idx:
    print(i, "blk", g["blk"][i], h["blk"][i], "ijk", g["ip"][i], g["jp"][i], h["ip"][i], h["jp"][i], "x", g["x"][i], h["x"][i], "y", g["y"][i], h["y"][i], "t", g["t"][i], h["t"][i])
same = ~diff
print("max |dx| among same:", np.abs(g["x"][same] - h["x"][same]).max())
