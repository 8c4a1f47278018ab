import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
from helpers import load_deck, make_oracle
from oracle import orc
from jaybenne_amd import mcblock
deck = sys.argv[1] if len(sys.argv) > 1 else "inf"
pin = load_deck(deck)
drv = mcblock.McblockDriver(pin, device=torch.device("cuda", 0))
O, _, _ = make_oracle(load_deck(deck), orc.MATH_PORTABLE, capacity_factor=40.0)
dt = pin.GetReal("jaybenne", "dt")
def cmp(tag):
    g = drv.md.get_swarm()
    og, oo = np.argsort(g["id"]), np.argsort(O.sw["id"][:O.n])
    print(tag, "n", drv.md.n, O.n)
    for k in ("id", "rng", "ip", "jp", "kp", "blk") + tuple(orc.SWARM_F64):
        a, b = g[k][og], O.sw[k][:O.n][oo]
        bad = np.nonzero(a != b)[0]
        if bad.size:
            print("  ", k, "mismatch", bad.size, "first ids", g["id"][og][bad[:4]], a[bad[:4]], b[bad[:4]], (a[bad[:4]] - b[bad[:4]]) if a.dtype == np.float64 else "")
cmp("init")
for cyc in range(14):
    drv.Step(); O.RadiationStep(cyc * dt, dt)
    cmp("cycle %d" % cyc)
