# one-off full-size check: C2 with 1e7 photons, 3 cycles, DefragParticles after every cycle against none:
# every photon (by creation id) bit-identical, tally equal to rounding (order of the atomic adds)
import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch, bench
from jaybenne_amd import mcblock
res = {}
for k in (0, 1):
    drv = mcblock.McblockDriver(bench.make_deck(1, 10_000_000, 64, "c2"), device=torch.device("cuda", 0), capacity_factor=1.5)
    drv.pkg.set_arithmetic("exact")
    drv.md.defrag_interval = k
    for _ in range(3):
        drv.Step()
    g = drv.md.get_swarm()
    o = np.argsort(g["id"], kind="stable")
    res[k] = ({n: v[o] for n, v in g.items() if n not in ("blk", "ip", "jp", "kp")}, drv.md.get_field("tally").copy(), drv.md.events)
    del drv
a, b = res[0], res[1]
print("events equal:", a[2] == b[2], a[2])
for n in a[0]:
    print(n, "identical" if np.array_equal(a[0][n], b[0][n]) else "DIFFERENT")
t0, t1 = a[1], b[1]
m = np.abs(t0) > 0
print("tally max rel diff:", float(np.max(np.abs(t0[m] - t1[m]) / np.abs(t0[m]))))
