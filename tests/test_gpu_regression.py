"""The reference's regression suite (tst/stepdiff.py, tst/stepdiff_smr.py and the deck list of
.github/workflows/ci.yml:122-140) on the GPU, through the same command-line surface."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from jaybenne_amd.deck import DECK_DIR as DECKS  # noqa: E402  (the reference's decks, data of the package)
STEPDIFF = ["parthenon/mesh/nx1=128", "parthenon/meshblock/nx1=128"]
SMR = ["parthenon/mesh/nx1=64", "parthenon/mesh/nx2=32", "parthenon/meshblock/nx1=16",
       "parthenon/meshblock/nx2=16"]


# (every deck in both arithmetic variants: "exact" -- JB_EXACT_ARITH=1, what the other tests run -- and
# "lean", the library's shipped default; the `lean` marker makes tests/conftest.py leave the
# default alone.  The all-DDMC decks take no lean step: one variant.)
@pytest.mark.parametrize("deck,overrides,tol", [
    ("stepdiff", STEPDIFF, 0.05), ("stepdiff_ddmc", STEPDIFF, 0.05),
    ("stepdiff_smr", SMR, 0.3), ("stepdiff_smr_ddmc", SMR, 0.3), ("stepdiff_smr_hybrid", SMR, 0.3),
    pytest.param("stepdiff", STEPDIFF, 0.05, marks=pytest.mark.lean, id="stepdiff-lean"),
    pytest.param("stepdiff_smr", SMR, 0.3, marks=pytest.mark.lean, id="stepdiff_smr-lean"),
    pytest.param("stepdiff_smr_hybrid", SMR, 0.3, marks=pytest.mark.lean, id="stepdiff_smr_hybrid-lean")])
def test_reference_regression_suite(gpu_device, deck, overrides, tol, capsys):
    from jaybenne_amd.__main__ import main
    rc = main(["-i", os.path.join(DECKS, deck + ".in"), "--tolerance", str(tol)] + overrides)
    out = capsys.readouterr().out
    assert rc == 0 and "TEST PASSED" in out, out


def test_cli_writes_the_decks_hdf5_dump(gpu_device, tmp_path, capsys):
    """inputs/stepdiff_smr.in:86-94 asks for an hdf5 dump of four fields and the swarm's x, y: the
    command line writes <problem_id>.out0.final.phdf, and the reference's acceptance loop run on
    the dump (cell centres from block bounds + Get, tst/regression_test.py:361-406) reproduces
    the number the command line printed."""
    from jaybenne_amd import analysis, phdf
    from jaybenne_amd.__main__ import main
    if not phdf.available():
        pytest.skip("libhdf5 not found")
    rc = main(["-i", os.path.join(DECKS, "stepdiff_smr.in"), "--tolerance", "0.3",
               "--output-dir", str(tmp_path)] + SMR)
    out = capsys.readouterr().out
    assert rc == 0, out
    d = phdf.read_dump(str(tmp_path / "stepdiff.out0.final.phdf"))
    assert d.NumDims == 2 and d.NumBlocks == 20 and sorted(set(d.Levels.tolist())) == [0, 1]
    assert d.Variables == ["field.material.density", "field.material.sie",
                           "field.material.internal_energy", "field.jaybenne.energy_tally"]
    var = d.Get("field.jaybenne.energy_tally")
    sol = analysis.ur_solution(d.Time, d.X1c)
    frac = np.abs(sol - var) / np.abs((sol + var) / 2.0)
    weighted = float((frac * sol).sum() / sol.sum())
    printed = float([ln for ln in out.splitlines() if ln.startswith("Mean weighted")][0].split()[-1])
    assert weighted == pytest.approx(printed, rel=2e-2)       # (printed with 3 digits)
    ph = d.GetSwarm("photons")
    assert ph.x is not None and ph.y is not None and ph.z is None and len(ph.x) == int(ph.counts.sum())
    assert np.all(d.Get("field.material.density") == 1.0)


def test_smr_noise_floor_with_more_particles(gpu_device, capsys):
    """The 0.3 gate of the SMR decks is Monte Carlo noise (20 particles per cell); with 20x the
    particles the same problem must come down by ~sqrt(20): a systematic error would not."""
    from jaybenne_amd.__main__ import main
    rc = main(["-i", os.path.join(DECKS, "stepdiff_smr_hybrid.in"), "--tolerance", "0.08",
               "jaybenne/num_particles=2000000"] + SMR)
    assert rc == 0, capsys.readouterr().out


def test_three_level_hybrid_extension(gpu_device, tmp_path, capsys):
    """BASELINE config C5: the shipped hybrid deck (coarse DDMC / fine IMC) with a nested level-2
    region added (sigma dx = 1.95, IMC) -- a synthetic extension pinned by the erf solution."""
    from jaybenne_amd.__main__ import main
    text = open(os.path.join(DECKS, "stepdiff_smr_hybrid.in")).read()
    text += ("<parthenon/static_refinement2>\nlevel = 2\nx1min = -0.125\nx1max = 0.125\n"
             "x2min = -0.125\nx2max = 0.125\nx3min = -0.25\nx3max = 0.25\n")
    deck = tmp_path / "hybrid3.in"
    deck.write_text(text)
    # 32 blocks x 1024 cells: 2e7 particles = 610 per cell, Monte Carlo noise ~ 5 %
    rc = main(["-i", str(deck), "--tolerance", "0.08", "jaybenne/num_particles=20000000"])
    out = capsys.readouterr().out
    assert "levels [0, 1, 2]" in out and rc == 0, out


@pytest.mark.parametrize("deck,tol", [("inf", 0.03), ("inf_stiff", 0.04),
                                      pytest.param("inf", 0.03, marks=pytest.mark.lean, id="inf-lean")])
def test_infinite_medium_decks(gpu_device, deck, tol):
    """inputs/inf.in and inputs/inf_stiff.in (100 and 10 cycles; the swarm pool grows as emission
    particles accumulate): the domain-mean radiation energy density stays at a T0^4.  inf runs as
    shipped; inf_stiff as shipped leaves ~35 census particles per cycle (5 % noise on the mean),
    so it runs with 16 x its particle count (~1 %)."""
    from helpers import load_deck
    from jaybenne_amd import constants, mcblock
    ov = {"jaybenne/num_particles": 160000} if deck == "inf_stiff" else None
    drv = mcblock.McblockDriver(load_deck(deck, ov), device=gpu_device)
    ur = 4.0 * constants.STEFAN_BOLTZMANN / constants.SPEED_OF_LIGHT * drv.mcb.initial_temperature ** 4
    sl = drv.mesh.interior()
    cap0, ratio = drv.md.capacity, []
    while drv.time < drv.tlim:
        drv.Step()
        ratio.append(float(drv.md.get_field("tally")[sl].mean()) / ur)
    assert abs(np.mean(ratio) - 1.0) < tol, ratio
    if deck == "inf":
        assert drv.ncycle in (100, 101) and drv.md.capacity > cap0   # the pool had to grow


def test_infinite_medium_cli(gpu_device, capsys):
    from jaybenne_amd.__main__ import main
    rc = main(["-i", os.path.join(DECKS, "inf.in"), "--tolerance", "0.2", "--comparison", "mean"])
    out = capsys.readouterr().out
    assert rc == 0 and "TEST PASSED" in out and "a T0^4" in out, out


def test_full_size_c2_profile(gpu_device, capsys):
    """BASELINE configs[1] at full size (256^3 cells in 64 blocks, 1e7 photons, 10 cycles): the
    energy-deposition profile, averaged over the transverse planes (0.6 photons per cell make a
    cell-by-cell comparison pure noise), against the reference's analytic solution and gate.
    With npc = 0.596 < 1 the reference's source rounding puts only npc of the energy on the mesh
    (sourcing.cpp:99-103), so the solution is scaled to the tally's total energy."""
    from jaybenne_amd.__main__ import main
    ov = ["jaybenne/num_particles=10000000"]
    for d in (1, 2, 3):
        ov += [f"parthenon/mesh/nx{d}=256", f"parthenon/meshblock/nx{d}=64"]
    rc = main(["-i", os.path.join(DECKS, "stepdiff.in"), "--tolerance", "0.05",
               "--transverse-average", "--match-total-energy"] + ov)
    out = capsys.readouterr().out
    assert rc == 0 and "TEST PASSED" in out, out


def test_full_size_c3_profile(gpu_device, capsys):
    """BASELINE configs[2] (stepdiff_ddmc, 3-D 128^3 cells in 8 blocks, every step DDMC) with 2e7
    photons, 10 cycles: plane-averaged profile against the analytic solution and the reference's
    stepdiff gate."""
    from jaybenne_amd.__main__ import main
    ov = ["jaybenne/num_particles=20000000"]
    for d in (1, 2, 3):
        ov += [f"parthenon/mesh/nx{d}=128", f"parthenon/meshblock/nx{d}=64"]
    rc = main(["-i", os.path.join(DECKS, "stepdiff_ddmc.in"), "--tolerance", "0.05",
               "--transverse-average"] + ov)
    out = capsys.readouterr().out
    assert rc == 0 and "TEST PASSED" in out, out


@pytest.mark.parametrize("deck,overrides", [
    ("stepdiff", ["parthenon/mesh/nx1=128", "parthenon/meshblock/nx1=64", "jaybenne/num_particles=20000"]),
    ("stepdiff_ddmc", ["parthenon/mesh/nx1=32", "parthenon/mesh/nx2=8", "parthenon/mesh/nx3=8",
                       "parthenon/meshblock/nx1=16", "parthenon/meshblock/nx2=4",
                       "parthenon/meshblock/nx3=4", "jaybenne/num_particles=30000"]),
    ("stepdiff_smr_hybrid", ["jaybenne/num_particles=30000", "parthenon/time/nlim=2"]),   # 2-D, 2 levels
    ("stepdiff", ["parthenon/mesh/nx1=64", "parthenon/meshblock/nx1=32", "jaybenne/num_particles=20000",
                  "mcblock/time_scale=2.0", "mcblock/mass_scale=3.0", "mcblock/length_scale=5.0",
                  "mcblock/temperature_scale=7.0", "mcblock/opacity_model=constant",
                  "mcblock/opacity_constant_value=0.5"]),       # non-unit code -> CGS scales, absorbing
    ("stepdiff_smr_ddmc", ["parthenon/mesh/nx1=32", "parthenon/mesh/nx2=16", "parthenon/mesh/nx3=16",
                           "parthenon/meshblock/nx1=8", "parthenon/meshblock/nx2=8",
                           "parthenon/meshblock/nx3=8", "jaybenne/num_particles=30000",
                           "parthenon/time/nlim=2"]),                                    # 3-D, 72 blocks
    ("stepdiff_smr_hybrid", ["jaybenne/num_particles=30000", "parthenon/time/nlim=4",
                             "jaybenne/defrag_interval=2"])])   # DefragParticles after cycles 2 and 4
def test_native_cpp_host_application(gpu_device, tmp_path, deck, overrides):
    """examples/mcblock_amd: the C++ host application on include/jaybenne_amd.hpp (deck parser,
    uniform mesh, device buffers through the HIP runtime, cycle loop) -- no Python, no PyTorch in
    that process.  Its particles and tally equal the Python driver's on the same deck."""
    import subprocess
    from helpers import ROOT, load_deck
    from jaybenne_amd import mcblock
    exe = os.path.join(ROOT, "examples", "mcblock_amd")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(ROOT, "examples")], check=True, capture_output=True)
    dump = tmp_path / "native.bin"
    res = subprocess.run([exe, "-i", os.path.join(DECKS, deck + ".in"), "--dump", str(dump)] + overrides,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    raw = dump.read_bytes()
    ncell, n, events = np.frombuffer(raw, dtype=np.int64, count=3)
    off = 24
    tally = np.frombuffer(raw, dtype=np.float64, count=ncell, offset=off); off += 8 * ncell
    ids = np.frombuffer(raw, dtype=np.uint64, count=n, offset=off); off += 8 * n
    xs = np.frombuffer(raw, dtype=np.float64, count=n, offset=off)
    drv = mcblock.McblockDriver(load_deck(deck, dict(o.split("=") for o in overrides)), device=gpu_device)
    drv.Execute()
    g = drv.md.get_swarm()
    assert n == drv.md.n and events == drv.md.events
    assert np.array_equal(np.sort(ids), np.sort(g["id"]))
    assert np.array_equal(xs[np.argsort(ids)], g["x"][np.argsort(g["id"])])
    want = drv.md.get_field("tally")[drv.mesh.interior()].ravel()
    np.testing.assert_allclose(tally, want, rtol=1e-12, atol=0)


@pytest.mark.parametrize("deck,overrides", [
    ("stepdiff", ["parthenon/mesh/nx1=16", "parthenon/meshblock/nx1=8", "jaybenne/num_particles=20000",
                  "jaybenne/do_emission=true", "jaybenne/do_feedback=true", "mcblock/opacity_model=constant",
                  "mcblock/opacity_constant_value=40.0", "mcblock/scattering_constant_value=20.0",
                  "mcblock/initial_temperature=1.0e6", "parthenon/time/nlim=3"]),
    ("stepdiff_smr_hybrid", ["parthenon/mesh/nx1=64", "parthenon/mesh/nx2=32", "parthenon/meshblock/nx1=16",
                             "parthenon/meshblock/nx2=16", "jaybenne/num_particles=30000",
                             "jaybenne/do_emission=true", "jaybenne/do_feedback=true",
                             "mcblock/opacity_model=constant", "mcblock/opacity_constant_value=20.0",
                             "parthenon/time/nlim=3"])])
def test_native_cpp_host_application_with_feedback(gpu_device, tmp_path, deck, overrides):
    """The native application's host update tasks (ghost-zone refresh of internal_energy through
    jb_fill_cells from its own source map, UpdateDerived as its own kernel) against the Python
    driver's: absorbing / emitting material with feedback, 3 cycles (1e-11 on attributes: from
    cycle 2 on u carries the order of the absorption atomics in its last bits)."""
    import subprocess
    from helpers import ROOT, load_deck
    from jaybenne_amd import mcblock
    exe = os.path.join(ROOT, "examples", "mcblock_amd")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(ROOT, "examples")], check=True, capture_output=True)
    dump = tmp_path / "native.bin"
    res = subprocess.run([exe, "-i", os.path.join(DECKS, deck + ".in"), "--dump", str(dump)] + overrides,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    raw = dump.read_bytes()
    ncell, n, events = np.frombuffer(raw, dtype=np.int64, count=3)
    off = 24
    tally = np.frombuffer(raw, dtype=np.float64, count=ncell, offset=off); off += 8 * ncell
    ids = np.frombuffer(raw, dtype=np.uint64, count=n, offset=off); off += 8 * n
    xs = np.frombuffer(raw, dtype=np.float64, count=n, offset=off)
    drv = mcblock.McblockDriver(load_deck(deck, dict(o.split("=") for o in overrides)), device=gpu_device,
                                capacity_factor=8.0)
    drv.Execute()
    g = drv.md.get_swarm()
    assert n == drv.md.n and events == drv.md.events
    assert np.array_equal(np.sort(ids), np.sort(g["id"]))
    np.testing.assert_allclose(xs[np.argsort(ids)], g["x"][np.argsort(g["id"])], rtol=1e-11, atol=0)
    want = drv.md.get_field("tally")[drv.mesh.interior()].ravel()
    np.testing.assert_allclose(tally, want, rtol=1e-11, atol=0)
