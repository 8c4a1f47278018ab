"""Larger and longer parity runs than the test suite affords (development aid; run on the GPU box):
HIP path (exact arithmetic) against the CPU oracle, photon by photon by creation id, on decks that reach
the rarer paths of the tracking kernels -- level changes and general relocations of DDMC photons, 1-D
reflections, hybrid interfaces -- with 10-50 times the photons of tests/test_gpu_parity.py."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("JB_EXACT_ARITH", "1")
import torch  # noqa: E402

from jaybenne_amd import mcblock  # noqa: E402
from jaybenne_amd.deck import load_deck  # noqa: E402
from oracle import orc  # noqa: E402
from oracle.harness import make_oracle, run_oracle_cycles  # noqa: E402

SMR = {"parthenon/mesh/nx1": 64, "parthenon/mesh/nx2": 32, "parthenon/meshblock/nx1": 16, "parthenon/meshblock/nx2": 16}
SMR3D = {"parthenon/mesh/nx1": 32, "parthenon/mesh/nx2": 16, "parthenon/mesh/nx3": 16,
         "parthenon/meshblock/nx1": 8, "parthenon/meshblock/nx2": 8, "parthenon/meshblock/nx3": 8}
LEVEL2 = ("<parthenon/static_refinement2>\nlevel = 2\nx1min = -0.125\nx1max = 0.125\n"
          "x2min = -0.125\nx2max = 0.125\nx3min = -0.25\nx3max = 0.25\n")
CASES = [
    ("2-D SMR, all DDMC", "stepdiff_smr_ddmc", dict(SMR, **{"jaybenne/num_particles": 1000000}), "", 5),
    ("3-D SMR, all DDMC (72 blocks)", "stepdiff_smr_ddmc", dict(SMR3D, **{"jaybenne/num_particles": 500000}), "", 3),
    ("1-D DDMC in 4 blocks", "stepdiff_ddmc", {"jaybenne/num_particles": 2000000, "parthenon/meshblock/nx1": 25}, "", 6),
    ("1-D DDMC, one block (records in LDS)", "stepdiff_ddmc", {"jaybenne/num_particles": 2000000, "parthenon/mesh/nx1": 128,
                                                              "parthenon/meshblock/nx1": 128}, "", 6),
    ("3-D uniform DDMC, 8 blocks of 16^3", "stepdiff_ddmc", {"jaybenne/num_particles": 1000000, "parthenon/mesh/nx1": 32,
                                                           "parthenon/mesh/nx2": 32, "parthenon/mesh/nx3": 32,
                                                           "parthenon/meshblock/nx1": 16, "parthenon/meshblock/nx2": 16,
                                                           "parthenon/meshblock/nx3": 16}, "", 4),
    ("3-level hybrid (configs[4])", "stepdiff_smr_hybrid", {"jaybenne/num_particles": 200000}, LEVEL2, 3),
    ("absorbing 2-D SMR DDMC", "stepdiff_smr_ddmc", dict(SMR, **{"jaybenne/num_particles": 400000, "mcblock/opacity_model": "constant",
                                                                "mcblock/opacity_constant_value": 20.0,
                                                                "jaybenne/do_emission": "true", "jaybenne/do_feedback": "false"}), "", 3),
]
IMC3D = {"parthenon/mesh/nx1": 32, "parthenon/mesh/nx2": 32, "parthenon/mesh/nx3": 32, "parthenon/meshblock/nx1": 16,
         "parthenon/meshblock/nx2": 16, "parthenon/meshblock/nx3": 16}
CASES += [
    ("3-D IMC, 8 blocks of 16^3 (x-space kernel)", "stepdiff", dict(IMC3D, **{"jaybenne/num_particles": 100000}), "", 2),
    ("2-D SMR IMC", "stepdiff_smr", dict(SMR, **{"jaybenne/num_particles": 100000}), "", 2),
    ("1-D IMC, 4 blocks, reflecting walls", "stepdiff", {"jaybenne/num_particles": 100000, "parthenon/meshblock/nx1": 25}, "", 3),
]
# (second round: the same decks with another seed -- other streams, other rare events)
if len(sys.argv) > 1:
    CASES = [(n + f", seed {sys.argv[1]}", d, dict(o, **{"jaybenne/seed": int(sys.argv[1])}), e, c) for n, d, o, e, c in CASES]
ok = True
for name, deck, ov, extra, cycles in CASES:
    def pin_():
        p = load_deck(deck, ov)
        if extra:
            p.load_string(extra)
        return p
    t0 = time.time()
    drv = mcblock.McblockDriver(pin_(), device=torch.device("cuda", 0), capacity_factor=4.0)
    for _ in range(cycles):
        drv.Step()
    g = drv.md.get_swarm()
    variant = drv.md.lib.jb_last_transport_variant(drv.md.handle).decode()
    t1 = time.time()
    O, mesh, _ = make_oracle(pin_(), orc.MATH_PORTABLE, threads=16, capacity_factor=4.0)
    run_oracle_cycles(O, pin_(), cycles)
    n = O.n
    og, oo = np.argsort(g["id"], kind="stable"), np.argsort(O.sw["id"][:n], kind="stable")
    bad = []
    if drv.md.n != n or drv.md.events != O.events:
        bad.append(f"n {drv.md.n} vs {n}, events {drv.md.events} vs {O.events}")
    else:
        for k in ("id", "x", "y", "z", "vx", "vy", "vz", "t", "w", "e", "ip", "jp", "kp", "blk", "rng", "status"):
            a, b = g[k][og], O.sw[k][:n][oo]
            if not np.array_equal(a, b):
                bad.append(f"{k}: {int((a != b).sum())} photons differ")
        sl = mesh.interior()
        ta, tb = drv.md.get_field("tally")[sl], O.fields["tally"][sl]
        if not np.allclose(ta, tb, rtol=1e-11, atol=0):
            bad.append("tally")
    print(f"{'ok  ' if not bad else 'FAIL'} {name}: {n} photons, {cycles} cycles, {O.events} events, {variant} "
          f"(gpu {t1 - t0:.1f} s, oracle {time.time() - t1:.1f} s) {'; '.join(bad)}", flush=True)
    ok = ok and not bad
    del drv, O
sys.exit(0 if ok else 1)
