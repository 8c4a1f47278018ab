"""The initial state and the package constants the oracle and the product SHARE (both build them with
jaybenne_amd.mcblock.Initialize / ProblemGenerator and jaybenne_amd.constants) against a fixture
worked out by hand from the reference's sources (tests/golden/initial_state.json: mcblock.cpp:78-82,
155-199; sourcing.cpp:68-103) -- the treatment tests/golden/smr_topology.json gives the mesh.  Cell
centres are formed here from the hand-written block lists, not by jaybenne_amd.mesh.  The oracle's
initial source is then held to the same numbers: photons per cell, weights, total energy."""
import json
import os

import numpy as np
import pytest

from helpers import load_deck, make_oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIX = json.load(open(os.path.join(GOLDEN, "initial_state.json")))
TOPO = json.load(open(os.path.join(GOLDEN, "smr_topology.json")))


def _blocks(case):
    """[(level, x centres of the interior cells [nx1], cell volume)] per block, by hand-written geometry."""
    f = FIX[case]
    if case == "stepdiff_1d_128":
        return [(0, -0.5 + (np.arange(128) + 0.5) / 128.0)]
    t = TOPO["stepdiff_smr"]
    out = []
    for lev, lx1, lx2 in t["blocks"]:
        ext = t["block_extent_level0"][0] / (1 << lev)
        out.append((lev, t["domain_min"][0] + lx1 * ext + (np.arange(32) + 0.5) * (ext / 32)))
    return out


def test_package_constants():
    from jaybenne_amd import constants, mcblock
    k = FIX["constants"]
    assert constants.SPEED_OF_LIGHT == k["speed_of_light"] and constants.STEFAN_BOLTZMANN == k["stefan_boltzmann"]
    pkg = mcblock.Initialize(load_deck("stepdiff"))
    assert pkg.eos.cv == k["cv"] == 1.0 / (k["gamma"] - 1.0)
    assert pkg.opacity.c == k["speed_of_light"] and pkg.opacity.sb == k["stefan_boltzmann"]
    assert pkg.opacity.kappa == 0.0 and pkg.scattering.kappa_s == 1.0e3 and pkg.scattering.apm == 1.0
    # the reference's analytic profile is normalised with ur0 = 7.5646e5 (tst/stepdiff.py:34)
    assert k["a_T0_4"] == pytest.approx(k["reference_ur0"], rel=2e-4)


@pytest.mark.parametrize("case", ["stepdiff_1d_128", "stepdiff_smr"])
def test_problem_generator_against_the_hand_computed_state(case):
    from jaybenne_amd import mcblock
    from jaybenne_amd.mesh import Mesh
    f = FIX[case]
    pin = load_deck(f["deck"], f["overrides"])
    mesh = Mesh.from_deck(pin)
    ic = mcblock.ProblemGenerator(mesh, mcblock.Initialize(pin))
    blocks = _blocks(case)
    assert mesh.nblocks == f["blocks"] == len(blocks) and mesh.ncell == f["cells_per_block"]
    sl = mesh.interior()
    rho, sie, u = (ic[k][sl] for k in ("rho", "sie", "u"))
    assert np.all(rho == f["density"])
    for b, (lev, xc) in enumerate(blocks):
        want = np.where(xc < 0.0, f["sie_hot"], f["sie_cold"])          # (a function of x1 only)
        assert np.array_equal(sie[b], np.broadcast_to(want, sie[b].shape)), b
        assert np.array_equal(u[b], sie[b] * f["density"]), b
    if case == "stepdiff_1d_128":
        assert int(np.argmax(blocks[0][1] >= 0.0)) == f["first_cold_cell"]


@pytest.mark.parametrize("case", ["stepdiff_1d_128", "stepdiff_smr"])
def test_initial_source_of_the_oracle_against_the_hand_computed_numbers(case):
    from oracle import orc
    f = FIX[case]
    O, mesh, _ = make_oracle(load_deck(f["deck"], f["overrides"]), orc.MATH_PORTABLE)
    blocks = _blocks(case)
    lo, hi = f["photons_per_cell"]
    assert lo == int(np.floor(f["npc"])) and hi == lo + 1
    sl = mesh.interior()
    num = O.fields["src_num"][sl]
    assert set(np.unique(num)) <= {float(lo), float(hi)}
    n = O.n
    assert n == int(num.sum())
    assert abs(n - f["num_particles"]) < 5.0 * np.sqrt(f["num_particles"])   # stochastic rounding of npc per cell
    frac = f["npc"] - lo
    assert abs((num == hi).mean() - frac) < 5.0 * np.sqrt(frac * (1 - frac) / num.size)
    # every photon's weight is erad / (photons of its cell), erad by temperature and cell volume
    w, blk, x = O.sw["w"][:n], O.sw["blk"][:n], O.sw["x"][:n]
    total = 0.0
    for b, (lev, xc) in enumerate(blocks):
        key = str(lev)
        eh = f["erad_hot"][key] if isinstance(f["erad_hot"], dict) else f["erad_hot"]
        ec = f["erad_cold"][key] if isinstance(f["erad_cold"], dict) else f["erad_cold"]
        sel = blk == b
        hot = x[sel] < 0.0
        allowed_hot = np.array([eh / lo, eh / hi])
        allowed_cold = np.array([ec / lo, ec / hi])
        close = lambda a, allowed: np.all(np.min(np.abs(a[:, None] / allowed[None, :] - 1.0), axis=1) <= 2e-15)
        assert close(w[sel][hot], allowed_hot), b
        assert close(w[sel][~hot], allowed_cold), b
        ncell_hot = int((xc < 0.0).sum()) * (len(xc) if mesh.ndim > 1 else 1)
        total += ncell_hot * eh + (mesh.ncell - ncell_hot) * ec
    assert float(w.sum()) == pytest.approx(total, rel=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["stepdiff_1d_128", "stepdiff_smr"])
def test_initial_source_of_the_hip_path_against_the_hand_computed_numbers(gpu_device, case):
    """... and the product's own initial source (k_source_count / k_source_fill through the C ABI)."""
    from jaybenne_amd import mcblock
    f = FIX[case]
    drv = mcblock.McblockDriver(load_deck(f["deck"], f["overrides"]), device=gpu_device)
    mesh, md = drv.mesh, drv.md
    blocks = _blocks(case)
    lo, hi = f["photons_per_cell"]
    sl = mesh.interior()
    num = md.get_field("src_num")[sl]
    assert set(np.unique(num)) <= {float(lo), float(hi)} and md.n == int(num.sum())
    g = md.get_swarm()
    w, blk, x = g["w"][:md.n], g["blk"][:md.n], g["x"][:md.n]
    total = 0.0
    for b, (lev, xc) in enumerate(blocks):
        key = str(lev)
        eh = f["erad_hot"][key] if isinstance(f["erad_hot"], dict) else f["erad_hot"]
        ec = f["erad_cold"][key] if isinstance(f["erad_cold"], dict) else f["erad_cold"]
        sel = blk == b
        hot = x[sel] < 0.0
        close = lambda a, allowed: np.all(np.min(np.abs(a[:, None] / allowed[None, :] - 1.0), axis=1) <= 2e-15)
        assert close(w[sel][hot], np.array([eh / lo, eh / hi])), b
        assert close(w[sel][~hot], np.array([ec / lo, ec / hi])), b
        ncell_hot = int((xc < 0.0).sum()) * (len(xc) if mesh.ndim > 1 else 1)
        total += ncell_hot * eh + (mesh.ncell - ncell_hot) * ec
    assert float(w.sum()) == pytest.approx(total, rel=1e-12)
