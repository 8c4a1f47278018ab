import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from helpers import load_deck
from jaybenne_amd import mcblock, analysis
n = int(sys.argv[1]); nyz = int(sys.argv[2])
ov = {"jaybenne/num_particles": n, "parthenon/mesh/nx1": 256, "parthenon/meshblock/nx1": 64}
for d in (2, 3):
    ov[f"parthenon/mesh/nx{d}"] = nyz; ov[f"parthenon/meshblock/nx{d}"] = min(nyz, 64)
drv = mcblock.McblockDriver(load_deck("stepdiff", ov), device=torch.device("cuda", 0))
print("blocks", drv.mesh.nblocks, "n", drv.md.n)
sl = drv.mesh.interior()
def prof():
    t = drv.md.get_field("tally")[sl]
    xs = np.stack([drv.mesh.cell_centers(b, 0)[sl[3]] for b in range(drv.mesh.nblocks)])
    key = np.round((xs - drv.mesh.gmin[0]) / drv.mesh.blk_dx[0, 0] - 0.5).astype(int)
    s = np.zeros(256); c = np.zeros(256)
    np.add.at(s, key.ravel(), t.sum(axis=(1, 2)).ravel()); np.add.at(c, key.ravel(), t.shape[1] * t.shape[2])
    return s / c
p0 = prof()
print("init", p0[[0, 64, 120, 127, 128, 135, 200]])
while drv.time < drv.tlim:
    drv.Step()
p = prof()
xc = -0.5 + (np.arange(256) + 0.5) / 256
sol = analysis.ur_solution(drv.time, xc)
for i in (0, 64, 100, 120, 127, 128, 135, 150, 200):
    print(i, xc[i], p[i], sol[i])
print(analysis.analytic_errors(drv.mesh, drv.md.get_field("tally"), drv.time, transverse_average=True))
print("energy", p.sum(), sol.sum(), p0.sum())
