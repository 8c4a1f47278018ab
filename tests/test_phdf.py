"""phdf-layout HDF5 dumps: written through libhdf5 with ctypes, re-read with the
minimal reader that exposes what reference analysis/jhdf.py:32-92 does, and used the way
tst/regression_test.py:361-381 uses a dump (cell centres from the block bounds + Get(variable))."""
import numpy as np
import pytest

from helpers import load_deck
from jaybenne_amd import analysis, phdf
from jaybenne_amd.mesh import Mesh

pytestmark = pytest.mark.skipif(not phdf.available(), reason="libhdf5 not found")


def _fake_state(mesh, seed=0):
    rng = np.random.default_rng(seed)
    tally = rng.random(mesh.field_shape)
    rho = np.ones(mesh.field_shape)
    n = 5000
    blk = rng.integers(0, mesh.nblocks, n)
    sw = {"blk": blk, "swarm.x": rng.random(n), "swarm.y": rng.random(n), "swarm.z": np.zeros(n),
          "id": np.arange(n, dtype=np.uint64)}
    return tally, rho, sw


@pytest.mark.parametrize("deck", ["stepdiff", "stepdiff_smr", "inf"])
def test_dump_round_trip(tmp_path, deck):
    mesh = Mesh.from_deck(load_deck(deck))
    tally, rho, sw = _fake_state(mesh)
    path = str(tmp_path / f"{deck}.out0.final.phdf")
    phdf.write_dump(path, mesh, time=3.3e-10, dt=3.3e-11, ncycle=10,
                    variables={"field.jaybenne.energy_tally": tally, "field.material.density": rho},
                    swarms={"photons": sw}, input_text="<parthenon/job>\nproblem_id = " + deck)
    d = phdf.read_dump(path)
    assert d.Time == 3.3e-10 and d.NCycle == 10 and d.NumDims == mesh.ndim
    assert d.NumBlocks == mesh.nblocks and list(d.MeshBlockSize) == list(mesh.nx)
    assert d.Variables == ["field.jaybenne.energy_tally", "field.material.density"]
    sl = mesh.interior()
    got = d.Get("field.jaybenne.energy_tally")
    assert got.shape == (mesh.nblocks, mesh.nx[2], mesh.nx[1], mesh.nx[0])
    assert np.array_equal(got, tally[sl])
    assert d.Get("no.such.variable") is None
    for b in (0, mesh.nblocks - 1):
        bb = d.BlockBounds[b]
        assert bb[0] == mesh.blk_xmin[b, 0]
        assert bb[1] == pytest.approx(mesh.blk_xmin[b, 0] + mesh.nx[0] * mesh.blk_dx[b, 0], rel=1e-15)
        # jhdf.py:65-73: cell centres from the block bounds
        dx1 = (bb[1] - bb[0]) / d.NX1
        np.testing.assert_allclose(bb[0] + (np.arange(d.NX1) + 0.5) * dx1,
                                   mesh.cell_centers(b, 0)[sl[3]], rtol=1e-14)
        np.testing.assert_allclose(d.X1c[b, 0, 0, :], mesh.cell_centers(b, 0)[sl[3]], rtol=1e-14)
    assert np.array_equal(d.Levels, np.asarray(mesh.blk_level))
    ph = d.GetSwarm("photons")
    assert int(ph.counts.sum()) == 5000 and np.array_equal(ph.counts, np.bincount(sw["blk"], minlength=mesh.nblocks))
    order = np.argsort(sw["blk"], kind="stable")
    assert np.array_equal(ph.x, sw["swarm.x"][order]) and np.array_equal(ph.y, sw["swarm.y"][order])
    assert np.array_equal(ph.Get("id").view(np.uint64), sw["id"][order])
    assert d.GetSwarm("electrons") is None
    d.close()


def test_reference_metric_from_a_dump(tmp_path):
    """The acceptance loop of tst/regression_test.py:361-406 run on a dump gives the same five
    numbers as run on the fields in memory."""
    mesh = Mesh.from_deck(load_deck("stepdiff_smr"))
    t = 3.335641e-10
    sl = mesh.interior()
    tally = np.zeros(mesh.field_shape)
    for b in range(mesh.nblocks):
        tally[b][:, :, :] = analysis.ur_solution(t, mesh.cell_centers(b, 0))[None, None, :]
    tally *= 1.0 + 0.05 * np.random.default_rng(1).standard_normal(tally.shape)
    want = analysis.analytic_errors(mesh, tally, t)
    path = str(tmp_path / "stepdiff.out0.final.phdf")
    phdf.write_dump(path, mesh, t, 3.3e-11, 10, {"field.jaybenne.energy_tally": tally})
    d = phdf.read_dump(path)
    var = d.Get("field.jaybenne.energy_tally")
    sol = analysis.ur_solution(d.Time, d.X1c)
    err = np.abs(sol - var)
    frac = err / np.abs((sol + var) / 2.0)
    assert float((frac * sol).sum() / sol.sum()) == pytest.approx(want["mean_frac_error_weighted"], rel=1e-12)
    assert float(err.max()) == pytest.approx(want["max_error"], rel=1e-12)
    d.close()
