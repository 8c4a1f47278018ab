"""The library's default arithmetic in the gray IMC tracking kernels ("lean", include/jaybenne_amd.h)
and its stated tolerance.

The exact variant performs the oracle's operations one for one and is held to it bit for bit
(tests/test_gpu_parity.py).  The lean variant replaces four of them by cheaper ones that agree to
4e-15 (relative) -- face distance as numerator times a once-refined reciprocal, time step as distance / c by
multiplication, position update as one fused multiply-add per axis, logarithm without its
compensated sum.  Such a difference moves a photon by ~1e-15 of its path and changes its history
only where it flips a comparison (which event comes first, which side of a nudge threshold), so
the tolerance stated and tested here is:

  after full radiation cycles, every particle has the same integer attributes (cell, block,
  status, random-stream state) as in the oracle / the exact variant, and every floating-point
  attribute within 1e-9 -- relative to the domain size for positions, to c for velocities, to the
  time step for times, to its own magnitude for weights; cell tallies within 1e-9 relative.

(`test_gpu_accuracy.py` states the tolerance against the reference's libm arithmetic.)"""
import numpy as np
import pytest

from helpers import load_deck, make_oracle, run_oracle_cycles
from test_gpu_parity import C5_LEVEL2, SMR3D, SMR_OVERRIDES, _gpu_problem

pytestmark = [pytest.mark.gpu, pytest.mark.lean]

C_LIGHT = 2.99792458e10

CASES = [
    ("stepdiff", {"jaybenne/num_particles": 4000}, 2),                       # 1-D, 2 blocks (as shipped)
    ("stepdiff", {"parthenon/mesh/nx1": 128, "parthenon/meshblock/nx1": 128,
                  "jaybenne/num_particles": 4000}, 2),                        # the reference's test shape
    ("stepdiff_smr", dict(SMR_OVERRIDES, **{"jaybenne/num_particles": 6000}), 1),      # 2-D SMR IMC
    ("stepdiff", {"parthenon/mesh/nx2": 8, "parthenon/mesh/nx3": 8, "parthenon/mesh/nx1": 16,
                  "parthenon/meshblock/nx1": 8, "parthenon/meshblock/nx2": 4,
                  "parthenon/meshblock/nx3": 4, "jaybenne/num_particles": 3000}, 1),   # 3-D, 8 blocks
    ("stepdiff", {"mcblock/opacity_model": "constant", "mcblock/opacity_constant_value": 40.0,
                  "mcblock/scattering_constant_value": 20.0, "mcblock/initial_temperature": 1.0e6,
                  "parthenon/mesh/nx1": 16, "parthenon/meshblock/nx1": 8,
                  "jaybenne/num_particles": 20000}, 1),                        # absorbing (GRAY = 1)
    ("stepdiff_smr_hybrid", {"jaybenne/num_particles": 30000}, 1),            # hybrid: IMC steps lean
    ("stepdiff_smr_hybrid", dict(C5_LEVEL2, **{"jaybenne/num_particles": 30000}), 1),   # ... 3 levels
    # 3-D SMR: level changes through the ghost-cell codes with four fine cells behind a coarse ghost cell
    ("stepdiff_smr", dict(SMR3D, **{"jaybenne/num_particles": 20000}), 1),             # pure IMC (k_imc_cell<3>)
    ("stepdiff_smr_hybrid", dict(SMR3D, **{"jaybenne/num_particles": 30000,
                                           "jaybenne/tau_ddmc": 20.0}), 1),            # coarse DDMC / fine IMC
    # cell widths that are not powers of two: the cell-local step does not care (only its conversions
    # to and from the swarm's coordinates round) -- uniform 3-D, and level changes in 2-D and 3-D SMR
    ("stepdiff_smr", {"parthenon/mesh/nx1": 60, "parthenon/mesh/nx2": 30, "parthenon/meshblock/nx1": 15,
                      "parthenon/meshblock/nx2": 15, "jaybenne/num_particles": 8000}, 1),
    ("stepdiff_smr", {"parthenon/mesh/nx1": 24, "parthenon/mesh/nx2": 12, "parthenon/mesh/nx3": 12,
                      "parthenon/meshblock/nx1": 6, "parthenon/meshblock/nx2": 6, "parthenon/meshblock/nx3": 6,
                      "jaybenne/num_particles": 20000}, 1),
    ("stepdiff", {"parthenon/mesh/nx1": 24, "parthenon/mesh/nx2": 12, "parthenon/mesh/nx3": 12,
                  "parthenon/meshblock/nx1": 12, "parthenon/meshblock/nx2": 6,
                  "parthenon/meshblock/nx3": 6, "jaybenne/num_particles": 4000}, 1),
    ("stepdiff_smr_hybrid", {"parthenon/mesh/nx1": 120, "parthenon/mesh/nx2": 60,
                             "parthenon/meshblock/nx1": 30, "parthenon/meshblock/nx2": 30,
                             "jaybenne/num_particles": 30000}, 1),   # (sigma dx = 8.3 coarse, 4.2 fine)
    # photons leaving through `outflow` swarm boundaries (status ESCAPED, absolute position kept): 1-D,
    # and 3-D with every face of the box open
    ("stepdiff", {"parthenon/swarm/ix1_bc": "outflow", "parthenon/swarm/ox1_bc": "outflow",
                  "jaybenne/num_particles": 20000, "mcblock/scattering_constant_value": 20.0}, 2),
    ("stepdiff", dict({f"parthenon/swarm/{s}x{a}_bc": "outflow" for s in "io" for a in "123"},
                      **{"parthenon/mesh/nx2": 8, "parthenon/mesh/nx3": 8, "parthenon/mesh/nx1": 16,
                         "parthenon/meshblock/nx1": 8, "parthenon/meshblock/nx2": 4,
                         "parthenon/meshblock/nx3": 4, "jaybenne/num_particles": 20000,
                         "mcblock/scattering_constant_value": 20.0}), 2),
]


def _close(a, b, scale, what, tol=1e-9):
    bad = np.abs(a - b) > tol * scale
    assert not bad.any(), (what, int(bad.sum()), a[bad][:3], b[bad][:3])


def _compare_within_tolerance(g, ref, n, mesh, dt, by_id=False, tol=1e-9):
    og = np.argsort(g["id"]) if by_id else slice(None)
    orf = np.argsort(ref["id"][:n]) if by_id else slice(None)
    for k in ("id", "rng", "ip", "jp", "kp", "blk", "status"):
        assert np.array_equal(g[k][og], ref[k][:n][orf]), k
    size = float(np.max(np.asarray(mesh.gmax) - np.asarray(mesh.gmin)))
    for k in ("x", "y", "z"):
        _close(g[k][og], ref[k][:n][orf], size, k, tol)
    for k in ("vx", "vy", "vz"):
        _close(g[k][og], ref[k][:n][orf], C_LIGHT, k, tol)
    _close(g["t"][og], ref["t"][:n][orf], dt, "t", tol)
    for k in ("w", "e"):
        _close(g[k][og], ref[k][:n][orf], np.abs(ref[k][:n][orf]), k, tol)


def test_lean_operations_against_the_exact_ones(gpu_device):
    """The two replaced operations with a rounding of their own: quotient as numerator times a
    once-refined reciprocal against the correctly rounded quotient (<= 2^-48 relative: 32 ulp;
    measured 19), logarithm without its compensated sum against the <= 1 ulp one (<= 3 ulp),
    square root with one residual correction instead of two (<= 2 ulp).  (The fused position update is the more
    accurate of the two forms.)"""
    import ctypes as C
    from jaybenne_amd import _lib
    from oracle import orc
    lib = _lib.load()
    p, e = _lib.Params(num_particles=10, dt=1.0), _lib.Eos(model=0, gm1=0.6, cv=1.5)
    o, s = _lib.Opacity(model=0, kappa=0.0, c=3e10, sb=5.67e-5), _lib.Scattering(model=0, kappa_s=1.0, apm=1.0)
    ctx = C.c_void_p()
    assert lib.jb_initialize(C.byref(p), C.byref(e), C.byref(o), C.byref(s), 0, C.byref(ctx)) == _lib.JB_COMPLETE
    rng = np.random.default_rng(3)
    n = 1 << 20
    x = (10.0 ** rng.uniform(-6, 12, n)) * rng.choice([-1.0, 1.0], n)
    got = np.empty(n)
    assert lib.jb_debug_math(ctx, 12, x.ctypes.data, n, got.ctypes.data) == _lib.JB_COMPLETE
    want = x / np.roll(x, -1)
    assert (np.abs(got - want) / np.spacing(np.abs(want))).max() <= 32.0
    u = np.concatenate([rng.random(n), 1.0 - 10.0 ** rng.uniform(-16, -1, n // 4), 10.0 ** rng.uniform(-300, 0, n // 4)])
    u = u[(u > 0) & (u < 1)]
    got = np.empty(u.size)
    assert lib.jb_debug_math(ctx, 13, u.ctypes.data, u.size, got.ctypes.data) == _lib.JB_COMPLETE
    orc.set_math_mode(orc.MATH_PORTABLE)
    want = orc.math_log(u)
    assert (np.abs(got - want) / np.spacing(np.abs(want))).max() <= 3.0
    # ... and on the 1024-row table the cell-local IMC kernel reads (series cut after r^5 / 5)
    assert lib.jb_debug_math(ctx, 15, u.ctypes.data, u.size, got.ctypes.data) == _lib.JB_COMPLETE
    err = (np.abs(got - want) / np.spacing(np.abs(want))).max()
    print("lean log, 1024 rows: largest error", err, "ulp")
    assert err <= 3.0
    # square root of 1 - mu^2 (scatter): one refinement of the hardware's reciprocal square root
    v = np.concatenate([1.0 - (2.0 * rng.random(n) - 1.0) ** 2, 2.0 ** rng.uniform(-52, 0, n // 4)])
    v = v[(v >= 2.0 ** -52) & (v <= 1.0)]
    got = np.empty(v.size)
    assert lib.jb_debug_math(ctx, 14, v.ctypes.data, v.size, got.ctypes.data) == _lib.JB_COMPLETE
    want = np.sqrt(v)
    err = (np.abs(got - want) / np.spacing(want)).max()
    print("lean sqrt: largest error", err, "ulp")
    assert err <= 2.0
    lib.jb_finalize(ctx)


@pytest.mark.parametrize("deck,overrides,cycles", CASES)
def test_lean_arithmetic_within_stated_tolerance_of_the_oracle(gpu_device, deck, overrides, cycles):
    from oracle import orc
    pin = load_deck(deck, overrides)
    drv = _gpu_problem(pin, gpu_device)
    assert drv.pkg.arithmetic() == "lean"                  # the library's default
    O, mesh, _ = make_oracle(load_deck(deck, overrides), orc.MATH_PORTABLE)
    for _ in range(cycles):
        drv.Step()
    run_oracle_cycles(O, pin, cycles)
    variant = drv.md.lib.jb_last_transport_variant(drv.md.handle).decode()
    if "hybrid" in deck:
        assert "k_hybrid" in variant and "lean" in variant, variant
    else:
        assert variant.endswith(("true>", "lean>")), variant   # k_transport<..., LEAN = true> / k_imc_cell<..., lean>
    assert drv.md.n == O.n and drv.md.events == O.events
    absorbing = "mcblock/opacity_constant_value" in overrides
    outflow = "parthenon/swarm/ix1_bc" in overrides
    if outflow:
        assert drv.md.stats()["n_escaped"] > 100
    _compare_within_tolerance(drv.md.get_swarm(), O.sw, O.n, mesh, pin.GetReal("jaybenne", "dt"),
                              by_id=absorbing or outflow, tol=1e-8 if cycles > 1 and outflow else 1e-9)
    sl = mesh.interior()
    a, b = drv.md.get_field("tally")[sl], O.fields["tally"][drv.md.gids][sl]
    assert np.abs(a - b).max() <= 1e-9 * np.abs(b).max()


def test_lean_and_exact_variants_agree_on_a_million_histories(gpu_device):
    """3-D, 8 blocks of 16^3, 1e6 photons, one cycle (1.4e9 events): the two variants of the kernel
    leave every photon in the same cell with the same stream state, attributes within the stated
    tolerance -- and are not the same arithmetic (some last bits differ)."""
    ov = {"parthenon/mesh/nx1": 32, "parthenon/mesh/nx2": 32, "parthenon/mesh/nx3": 32,
          "parthenon/meshblock/nx1": 16, "parthenon/meshblock/nx2": 16, "parthenon/meshblock/nx3": 16,
          "jaybenne/num_particles": 1000000}
    out = {}
    for mode in ("lean", "exact"):
        drv = _gpu_problem(load_deck("stepdiff", ov), gpu_device)
        drv.pkg.set_arithmetic(mode)
        drv.Step()
        assert drv.md.lib.jb_last_transport_variant(drv.md.handle).decode().endswith(
            ("true>", "lean>") if mode == "lean" else "false>")
        out[mode] = (drv.md.get_swarm(), drv.md.n, drv.md.events, drv.mesh, drv.md.get_field("tally"))
        del drv
    (g, n, ev, mesh, tl), (h, m, ev2, _, te) = out["lean"], out["exact"]
    assert n == m and ev == ev2
    _compare_within_tolerance(g, h, n, mesh, 3.335641e-11)
    assert np.any(g["x"] != h["x"])
    sl = mesh.interior()
    assert np.abs(tl[sl] - te[sl]).max() <= 1e-9 * np.abs(te[sl]).max()


def test_lean_against_exact_over_ten_cycles(gpu_device):
    """How the difference between the two variants grows with the number of cycles (the reference's
    own test runs 10: tst/stepdiff.py).  Every scatter turns a position difference into a
    direction-dependent path difference, so two roundings of the same history separate like any
    two nearby trajectories of a chaotic system -- measured on this deck (3-D, 1e5 photons), largest
    position difference over all photons, relative to the domain: 5e-12 after one cycle, 1.3e-9
    after two, then about a decade per cycle, 1e-4 after ten; 20 of the 1e5 photons had a
    comparison flipped on the way (a different sequence of events from there on).  (Round 3's
    x-space lean step: 5e-13 / 7e-11 / 7 photons -- it rounded x to the same absolute grid as the
    exact variant at every event; the cell-local step carries more bits than either.)  The oracle's own
    two flavours (libm / portable: two correct statements of the reference, <= 1 ulp apart in log
    and sincos) separate at the same rate.  Stated and asserted (include/jaybenne_amd.h): attributes
    within 1e-9 after one cycle, 1e-8 after two; after 10 cycles every photon whose event sequence was not flipped
    sits in the same cell as its exact twin with positions within 1e-2 of the domain, the flipped
    ones are < 1e-3 of the photons, and the energy tally -- what the reference's acceptance test
    reads -- is within 1e-9 of its largest value plus the weight of the photons that changed cell."""
    ov = {"parthenon/mesh/nx1": 32, "parthenon/mesh/nx2": 32, "parthenon/mesh/nx3": 32,
          "parthenon/meshblock/nx1": 16, "parthenon/meshblock/nx2": 16, "parthenon/meshblock/nx3": 16,
          "jaybenne/num_particles": 100000}
    out = {}
    for mode in ("lean", "exact"):
        drv = _gpu_problem(load_deck("stepdiff", ov), gpu_device)
        drv.pkg.set_arithmetic(mode)
        growth = []
        for cyc in range(10):
            drv.Step()
            growth.append(drv.md.get_swarm()["x"].copy())
        out[mode] = (drv.md.get_swarm(), drv.md.n, drv.mesh, drv.md.get_field("tally"), growth)
        del drv
    (g, n, mesh, tl, gl), (h, m, _, te, ge) = out["lean"], out["exact"]
    assert n == m
    same = g["rng"] == h["rng"]                      # event sequence not flipped
    frac_flipped = 1.0 - same.mean()
    print("photons whose event sequence differs after 10 cycles:", int((~same).sum()), "of", n)
    assert frac_flipped <= 1e-3
    for k in ("ip", "jp", "kp", "blk", "status"):
        assert np.array_equal(g[k][same], h[k][same]), k
    size = float(np.max(np.asarray(mesh.gmax) - np.asarray(mesh.gmin)))
    drift = [float(np.abs(a - b)[same].max() / size) for a, b in zip(gl, ge)]
    print("largest position difference / domain size after cycles 1..10:", ["%.1e" % d for d in drift])
    assert drift[0] <= 1e-9 and drift[1] <= 1e-8 and drift[-1] <= 1e-2
    sl = mesh.interior()
    w = float(h["w"].max())
    dv = float(mesh.cell_volume(0))
    moved = (~same) & ((g["ip"] != h["ip"]) | (g["jp"] != h["jp"]) | (g["kp"] != h["kp"]) | (g["blk"] != h["blk"]))
    allow = 1e-9 * np.abs(te[sl]).max() + 2.0 * w / dv * max(int(moved.sum()), 0)
    assert np.abs(tl[sl] - te[sl]).max() <= allow


def test_wide_flat_block_does_not_run_the_cell_local_kernel(gpu_device):
    """A 3-D block with ni * nj >= 2^20 (1020 x 1020 x 4 cells): the byte stride 8 ni nj of the cell-local
    step no longer fits the 24-bit multiply-add that forms the photon's cell offset, so
    jb_mesh_create keeps such a mesh on the x-space lean kernel -- same tolerance against the oracle."""
    from oracle import orc
    ov = {"parthenon/mesh/nx1": 1020, "parthenon/mesh/nx2": 1020, "parthenon/mesh/nx3": 4,
          "parthenon/meshblock/nx1": 1020, "parthenon/meshblock/nx2": 1020, "parthenon/meshblock/nx3": 4,
          "jaybenne/num_particles": 20000}
    pin = load_deck("stepdiff", ov)
    drv = _gpu_problem(pin, gpu_device)
    assert drv.pkg.arithmetic() == "lean"
    O, mesh, _ = make_oracle(load_deck("stepdiff", ov), orc.MATH_PORTABLE)
    drv.Step()
    run_oracle_cycles(O, pin, 1)
    variant = drv.md.lib.jb_last_transport_variant(drv.md.handle).decode()
    assert "k_imc_cell" not in variant and variant.endswith("true>"), variant
    assert drv.md.n == O.n and drv.md.events == O.events
    _compare_within_tolerance(drv.md.get_swarm(), O.sw, O.n, mesh, pin.GetReal("jaybenne", "dt"))
