"""Parity of the HIP history loop with the CPU oracle, through the C ABI, on a real MI355X.

Bar: random words and indices bit-exact; every floating-point particle attribute bit-exact against
the oracle in its portable-math flavour (the HIP code implements the same IEEE sequence); cell
tallies to 1e-12 relative (atomic accumulation order).  The oracle's libm flavour -- the
reference's own arithmetic -- is tied to the portable one in tests/test_oracle_math.py.
"""
import ctypes as C

import numpy as np
import pytest

from helpers import load_deck, make_oracle, run_oracle_cycles
from step_cases import step_cases

pytestmark = pytest.mark.gpu

SMR_OVERRIDES = {"parthenon/mesh/nx1": 64, "parthenon/mesh/nx2": 32,
                 "parthenon/meshblock/nx1": 16, "parthenon/meshblock/nx2": 16}


SMR3D = {"parthenon/mesh/nx1": 32, "parthenon/mesh/nx2": 16, "parthenon/mesh/nx3": 16,
         "parthenon/meshblock/nx1": 8, "parthenon/meshblock/nx2": 8, "parthenon/meshblock/nx3": 8}


@pytest.fixture(scope="module")
def ctx(gpu_device):
    """A bare package context for the debug entry points."""
    from jaybenne_amd import jaybenne as jb, mcblock
    pin = load_deck("stepdiff")
    mcb = mcblock.Initialize(pin)
    pkg = jb.Initialize(pin, mcb.opacity, mcb.scattering, mcb.eos, device=gpu_device)
    yield pkg
    pkg.close()


# ------------------------------------------------------------------------------------------------
def test_philox_known_answers_on_device(ctx):
    from jaybenne_amd import _lib
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        out = (C.c_uint32 * 4)()
        _lib.check(ctx.lib.jb_debug_philox(ctx.ctx, (C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out))
        assert tuple(out) == want


def test_stream_layout_matches_rocrand(ctx):
    """key = seed, counter = {block, 0, id_lo, id_hi} is rocrand_init(seed, subsequence=id, 0)."""
    from jaybenne_amd import _lib
    seed, sub = 349857, (7 << 32) | 123456789
    ref = (C.c_uint32 * 8)()
    _lib.check(ctx.lib.jb_debug_rocrand_philox(ctx.ctx, seed, sub, ref))
    for blk in (0, 1):
        out = (C.c_uint32 * 4)()
        ctr = (C.c_uint32 * 4)(blk, 0, sub & 0xffffffff, sub >> 32)
        key = (C.c_uint32 * 2)(seed, 0)
        _lib.check(ctx.lib.jb_debug_philox(ctx.ctx, ctr, key, out))
        assert list(out) == list(ref[4 * blk:4 * blk + 4])


def test_uniform_stream_bit_exact(ctx):
    """Philox seeding of a stream and the LCG draws, device vs oracle."""
    from jaybenne_amd import _lib
    from oracle import orc
    for domain, sid in ((0, 0), (0, 0x1234567890), (1, (3 << 44) | (17 << 24) | 4095)):
        st = C.c_uint64(0)
        _lib.check(ctx.lib.jb_debug_seed_state(ctx.ctx, 349857, domain, sid, C.byref(st)))
        assert st.value == orc.seed_state(349857, domain, sid)
        out = np.empty(257)
        fin = C.c_uint64(0)
        _lib.check(ctx.lib.jb_debug_draw_stream(ctx.ctx, st.value, out.size, out.ctypes.data,
                                                C.byref(fin)))
        want, want_fin = orc.draw_stream(st.value, out.size)
        assert np.array_equal(out, want) and fin.value == want_fin
        assert out.min() > 0.0 and out.max() < 1.0
    # particle streams: disjoint strided segments (csrc/jb_rng.hpp), device vs oracle
    for pid in (0, 1, 2, 12345, 2 ** 30 - 1, 2 ** 30, (3 << 30) | 777, 2 ** 40 + 99):
        st = C.c_uint64(0)
        _lib.check(ctx.lib.jb_debug_stream_start(ctx.ctx, 349857, pid, C.byref(st)))
        assert st.value == orc.stream_start(349857, pid), pid


def _dev_math(ctx, which, x):
    from jaybenne_amd import _lib
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    _lib.check(ctx.lib.jb_debug_math(ctx.ctx, which, x.ctypes.data, x.size, out.ctypes.data))
    return out


def test_device_math_bit_exact(ctx):
    from oracle import orc
    rng = np.random.default_rng(1)
    orc.set_math_mode(orc.MATH_PORTABLE)
    u = np.concatenate([rng.random(200000), 2.0 ** -rng.integers(1, 53, 1000) * rng.random(1000),
                        [2.0 ** -53, 1 - 2.0 ** -53, 0.5, 0.70710678, 0.75]])
    u = u[(u > 0) & (u < 1)]
    assert np.array_equal(_dev_math(ctx, 0, u), orc.math_log(u))
    prod = u[:50000] * u[50000:100000] * u[100000:150000] * u[150000:200000]   # Planck argument
    assert np.array_equal(_dev_math(ctx, 0, prod), orc.math_log(prod))
    phi = 2.0 * np.pi * u
    s, c = orc.math_sincos(phi)
    assert np.array_equal(_dev_math(ctx, 1, phi), s)
    assert np.array_equal(_dev_math(ctx, 2, phi), c)
    s2, c2 = orc.math_sincos2pi(u)
    assert np.array_equal(_dev_math(ctx, 9, u), s2)
    assert np.array_equal(_dev_math(ctx, 10, u), c2)
    mu = 2.0 * u - 1.0
    assert np.array_equal(_dev_math(ctx, 3, mu), orc.math_acos(mu))
    # IEEE sqrt and divide on the device are correctly rounded (the host's are)
    assert np.array_equal(_dev_math(ctx, 4, u), np.sqrt(u))
    assert np.array_equal(_dev_math(ctx, 5, u), 1.0 / u)
    # the range-restricted sqrt and divide of jb_math.hpp are the same correctly rounded results
    assert np.array_equal(_dev_math(ctx, 6, u), np.sqrt(u))
    for scale_a, scale_b in ((1.0, 1.0), (3.0e8, 3.0e10), (1.0e-9, 1.0e-12), (40.0, 3.0e13)):
        x = u[:100001].copy()
        x[0::2] *= scale_a      # numerators: distances x speed, -log(xi)
        x[1::2] *= scale_b      # denominators: velocity components, c x opacity
        want = x / np.roll(x, -1)
        assert np.array_equal(_dev_math(ctx, 7, x), want), (scale_a, scale_b)
    d = np.concatenate([u * 1.0e-2, u * 1.0e4, [0.0]])
    assert np.array_equal(_dev_math(ctx, 8, d), d / 2.99792458e10)


# ------------------------------------------------------------------------------------------------
def test_step_functions_bit_exact(ctx):
    from jaybenne_amd import _lib
    from oracle import orc
    orc.set_math_mode(orc.MATH_PORTABLE)
    which = {"transport": 0, "ddmc": 1, "albedo": 2}
    names = [n for n, _ in orc.Step._fields_]
    seen = set()
    for kind, d, tape in step_cases():
        so, sd = orc.Step(), _lib.DebugStep()
        for k, v in d.items():
            setattr(so, k, v)
            setattr(sd, k, v)
        n_o = orc.call_step(kind, so, tape)
        tp = np.ascontiguousarray(tape, dtype=np.float64)
        nd = C.c_int(0)
        _lib.check(ctx.lib.jb_debug_step_call(ctx.ctx, which[kind], C.byref(sd), tp.ctypes.data,
                                              tp.size, C.byref(nd)))
        assert nd.value == n_o, (kind, d, tape)
        for nm in names:
            a, b = getattr(so, nm), getattr(sd, nm)
            assert (a == b) or (a != a and b != b), (kind, nm, a, b, d, tape)
        seen.add((kind, so.is_absorbed, so.is_scattered, so.is_rejected, n_o))
    assert len(seen) >= 9         # the cases really do spread over the branches


def test_hand_written_step_vectors_on_the_device(ctx):
    """tests/golden/hand_step_vectors.json (worked out on paper from the reference's text) and SURVEY.md
    Appendix D's vector (computed by the reference's own header) through the device's step functions --
    pins that do not come from the oracle (tests/test_oracle_steps.py holds the oracle to the same file)."""
    from jaybenne_amd import _lib
    from test_oracle_steps import check_hand_case, hand_vectors
    which = {"transport": 0, "ddmc": 1, "albedo": 2}
    cases, app_d = hand_vectors()

    def run(kind, d, tape):
        sd = _lib.DebugStep()
        for k, v in d.items():
            setattr(sd, k, v)
        tp = np.ascontiguousarray(tape, dtype=np.float64)
        nd = C.c_int(0)
        _lib.check(ctx.lib.jb_debug_step_call(ctx.ctx, which[kind], C.byref(sd), tp.ctypes.data, tp.size,
                                              C.byref(nd)))
        return sd, nd.value

    for name, kind, d, tape, ndraws, exact, close in cases:
        sd, n = run(kind, d, tape)
        check_hand_case(name, sd, n, ndraws, exact, close)
    sd, n = run(app_d["kind"], app_d["in"], app_d["tape"])
    assert n == app_d["ndraws"]
    for k, v in app_d["expect"].items():
        assert getattr(sd, k) == v, (k, getattr(sd, k), v)


def test_sampling_functions_bit_exact(ctx):
    from jaybenne_amd import _lib
    from oracle import orc
    orc.set_math_mode(orc.MATH_PORTABLE)
    c = 2.99792458e10
    rng = np.random.default_rng(5)

    def dev(which, a, iv, tape):
        a8 = np.zeros(8); a8[:len(a)] = a
        i4 = np.zeros(4, dtype=np.int32); i4[:len(iv)] = iv
        tp = np.ascontiguousarray(tape, dtype=np.float64)
        out = np.zeros(4); io = np.zeros(2, dtype=np.int32); nd = C.c_int(0)
        _lib.check(ctx.lib.jb_debug_sample_call(ctx.ctx, which, a8.ctypes.data, i4.ctypes.data,
                                                tp.ctypes.data, tp.size, out.ctypes.data,
                                                io.ctypes.data, C.byref(nd)))
        return out, io, nd.value

    for _ in range(50):
        tape = rng.random(6)
        v, n = orc.call_scatter(c, tape)
        out, _, nd = dev(0, [c], [], tape)
        assert nd == n == 2 and np.array_equal(out[:3], v)
        for sgn in (1.0, -1.0):
            v, n = orc.call_face_iso_dir(sgn * c, tape)
            out, _, nd = dev(1, [sgn * c], [], tape)
            assert nd == n == 2 and np.array_equal(out[:3], v)
        e, n = orc.call_planck(5.670373e-5, 1.0e5, tape)
        out, _, nd = dev(2, [5.670373e-5, 1.0e5], [], tape)
        assert nd == n == 5 and out[0] == e
        i, x, n = orc.call_face_2d(7, 0.01, 0.3, 0.5, tape, 8, 0.25)
        out, io, nd = dev(3, [0.01, 0.3, 0.5, 0.25], [7, 8], tape)
        assert nd == n == 2 and io[0] == i and out[0] == x
        P = rng.random(4)
        ij, x12, n = orc.call_face_3d(3, 9, 0.01, 0.02, P, tape, [4, 10], [0.5, -0.25])
        out, io, nd = dev(4, [0.01, 0.02, *P, 0.5, -0.25], [3, 9, 4, 10], tape)
        assert nd == n == 3 and list(io) == ij and np.array_equal(out[:2], x12)
    # Planck series index > 1 (xi0 close to 1)
    e, n = orc.call_planck(5.670373e-5, 300.0, [0.9999, 0.5, 0.5, 0.5, 0.5])
    out, _, nd = dev(2, [5.670373e-5, 300.0], [], [0.9999, 0.5, 0.5, 0.5, 0.5])
    assert out[0] == e and nd == n


# ------------------------------------------------------------------------------------------------
def _compare_swarm(md, O, exact=True):
    from oracle import orc
    g = md.get_swarm()
    n = O.n
    assert md.n == n
    for k in ("id", "rng", "ip", "jp", "kp", "blk", "status"):
        assert np.array_equal(g[k], O.sw[k][:n]), k
    for k in orc.SWARM_F64:
        a, b = g[k], O.sw[k][:n]
        if exact:
            bad = np.nonzero(a != b)[0]
            assert bad.size == 0, (k, bad[:5], a[bad[:5]], b[bad[:5]])
        else:
            np.testing.assert_allclose(a, b, rtol=1e-9, atol=0, err_msg=k)


def _compare_swarm_by_id(md, O, exact=True):
    """Compaction reorders the survivors: compare as sets keyed by stream id."""
    from oracle import orc
    g = md.get_swarm()
    order_g = np.argsort(g["id"])
    order_o = np.argsort(O.sw["id"][:O.n])
    assert md.n == O.n
    for k in ("id", "rng", "ip", "jp", "kp", "blk"):
        assert np.array_equal(g[k][order_g], O.sw[k][:O.n][order_o]), k
    for k in orc.SWARM_F64:
        a, b = g[k][order_g], O.sw[k][:O.n][order_o]
        if exact:
            assert np.array_equal(a, b), k
        else:
            np.testing.assert_allclose(a, b, rtol=1e-11, atol=0, err_msg=k)


def _compare_fields(md, O, names=("tally", "edelta", "fleck", "src_num", "src_ew"), scale=None):
    """1e-12 relative; `scale` (per cell) is the magnitude of the terms summed into a cell when
    the sum itself cancels (energy_delta = absorbed - emitted)."""
    sl = md.mesh.interior()
    for k in names:
        a = md.get_field(k)[sl]
        b = O.fields[k][md.gids][sl]
        fin = np.isfinite(b)
        assert np.array_equal(np.isfinite(a), fin), k
        ref = np.abs(b) if scale is None else np.maximum(np.abs(b), scale)
        with np.errstate(invalid="ignore"):      # inf - inf where src_ew is inf on both sides
            bad = np.abs(a - b)[fin] > 1e-12 * ref[fin]
        assert not bad.any(), (k, a[fin][bad][:4], b[fin][bad][:4])


def _gpu_problem(pin, gpu_device):
    from jaybenne_amd import mcblock
    return mcblock.McblockDriver(pin, device=gpu_device)


# BASELINE configs[4]: the shipped hybrid deck plus a nested level-2 region (bench.make_deck "c5";
# the block list is pinned by hand in tests/golden/smr_topology.json): 3 levels, 32 blocks
C5_LEVEL2 = {"parthenon/static_refinement2/level": 2,
             "parthenon/static_refinement2/x1min": -0.125, "parthenon/static_refinement2/x1max": 0.125,
             "parthenon/static_refinement2/x2min": -0.125, "parthenon/static_refinement2/x2max": 0.125,
             "parthenon/static_refinement2/x3min": -0.25, "parthenon/static_refinement2/x3max": 0.25}

CASES = [
    # deck, overrides, cycles
    ("stepdiff", {"jaybenne/num_particles": 4000}, 2),                       # 1-D, 2 blocks (as shipped)
    ("stepdiff", {"parthenon/mesh/nx1": 128, "parthenon/meshblock/nx1": 128,
                  "jaybenne/num_particles": 4000}, 1),                        # the reference's test shape
    ("stepdiff_ddmc", {"jaybenne/num_particles": 20000}, 2),                  # 1-D all-DDMC
    ("stepdiff_smr", dict(SMR_OVERRIDES, **{"jaybenne/num_particles": 6000}), 1),      # 2-D SMR IMC
    ("stepdiff_smr_ddmc", dict(SMR_OVERRIDES, **{"jaybenne/num_particles": 40000}), 2),  # 2-D SMR DDMC
    ("stepdiff_smr_hybrid", {"jaybenne/num_particles": 30000}, 1),            # true IMC/DDMC hybrid
    ("stepdiff_smr_hybrid", dict(C5_LEVEL2, **{"jaybenne/num_particles": 30000}), 2),   # ... on the
    # 3-level mesh of BASELINE configs[4] (level 0 DDMC, levels 1 and 2 IMC)
    ("stepdiff", {"parthenon/mesh/nx2": 8, "parthenon/mesh/nx3": 8, "parthenon/mesh/nx1": 16,
                  "parthenon/meshblock/nx1": 8, "parthenon/meshblock/nx2": 4,
                  "parthenon/meshblock/nx3": 4, "jaybenne/num_particles": 3000}, 1),   # 3-D, 8 blocks
    ("stepdiff_ddmc", {"parthenon/mesh/nx2": 16, "parthenon/mesh/nx3": 16, "parthenon/mesh/nx1": 128,
                       "parthenon/meshblock/nx1": 32, "parthenon/meshblock/nx2": 8,
                       "parthenon/meshblock/nx3": 8, "jaybenne/num_particles": 40000,
                       "jaybenne/tau_ddmc": 5.0}, 2),                          # 3-D DDMC, 16 blocks
    ("stepdiff_smr_ddmc", dict(SMR3D, **{"jaybenne/num_particles": 40000}), 2),   # 3-D SMR (72 blocks,
    # 2 levels), all DDMC: coarse -> fine crossings pick one of 4 fine faces (SampleFace3D)
    ("stepdiff_smr_hybrid", dict(SMR3D, **{"jaybenne/num_particles": 30000,
                                           "jaybenne/tau_ddmc": 20.0}), 1),      # 3-D SMR, coarse DDMC / fine IMC
    # cell widths that are not powers of two: the general-geometry kernels (EXACT = false, k_hybrid
    # MODE 0 / 1), in 3-D pure IMC and on the 2-D hybrid deck
    ("stepdiff", {"parthenon/mesh/nx1": 24, "parthenon/mesh/nx2": 12, "parthenon/mesh/nx3": 12,
                  "parthenon/meshblock/nx1": 12, "parthenon/meshblock/nx2": 6,
                  "parthenon/meshblock/nx3": 6, "jaybenne/num_particles": 4000}, 1),
    ("stepdiff_smr_hybrid", {"parthenon/mesh/nx1": 120, "parthenon/mesh/nx2": 60,
                             "parthenon/meshblock/nx1": 30, "parthenon/meshblock/nx2": 30,
                             "jaybenne/num_particles": 30000}, 1),   # (sigma dx = 8.3 coarse, 4.2 fine)
]


@pytest.mark.parametrize("deck,overrides,cycles", CASES)
def test_history_loop_bit_exact(gpu_device, deck, overrides, cycles):
    from oracle import orc
    pin = load_deck(deck, overrides)
    drv = _gpu_problem(pin, gpu_device)
    O, mesh, _ = make_oracle(load_deck(deck, overrides), orc.MATH_PORTABLE)
    _compare_swarm(drv.md, O)          # sourcing
    _compare_fields(drv.md, O, ("tally", "src_num"))
    for _ in range(cycles):
        drv.Step()
    run_oracle_cycles(O, pin, cycles)
    _compare_swarm(drv.md, O)
    _compare_fields(drv.md, O)
    assert drv.md.events == O.events
    # every particle stops exactly at census, |v| = c or 0
    g = drv.md.get_swarm()
    assert np.all(g["t"] >= drv.time)


@pytest.mark.parametrize("deck,overrides,cycles", [c for c in CASES if c[0].endswith("_ddmc")])
def test_general_kernel_on_all_ddmc_meshes(gpu_device, deck, overrides, cycles, monkeypatch):
    """All-DDMC meshes are tracked by k_ddmc_all (test_history_loop_bit_exact runs it); k_hybrid,
    the kernel of a mesh that mixes IMC and DDMC cells, must give the same bits on them."""
    from oracle import orc
    monkeypatch.setenv("JB_NO_DDMC_ALL", "1")
    pin = load_deck(deck, overrides)
    drv = _gpu_problem(pin, gpu_device)
    O, mesh, _ = make_oracle(load_deck(deck, overrides), orc.MATH_PORTABLE)
    for _ in range(cycles):
        drv.Step()
    run_oracle_cycles(O, pin, cycles)
    assert "k_hybrid<" in drv.md.lib.jb_last_transport_variant(drv.md.handle).decode()
    _compare_swarm(drv.md, O)
    _compare_fields(drv.md, O)
    assert drv.md.events == O.events


@pytest.mark.parametrize("form", ["1", "2"])
@pytest.mark.parametrize("deck,overrides,cycles", [c for c in CASES if c[0].endswith("_ddmc")])
def test_quad_cooperative_gather_on_all_ddmc_meshes(gpu_device, deck, overrides, cycles, form, monkeypatch):
    """k_ddmc_all fetches its step records either with four loads per lane or, once the records of
    the resident blocks exceed 1 MiB (BASELINE configs[2] in 3-D: 160 MB), with the four lanes of a
    quad sharing the fetch of each record through LDS.  The test decks are far below that size:
    force the second form on them -- same bits -- with 32-bit byte offsets ("1") and with the 64-bit
    addresses it uses once the records span 4 GiB or more ("2")."""
    from oracle import orc
    monkeypatch.setenv("JB_COOP_GATHER", form)
    pin = load_deck(deck, overrides)
    drv = _gpu_problem(pin, gpu_device)
    O, mesh, _ = make_oracle(load_deck(deck, overrides), orc.MATH_PORTABLE)
    for _ in range(cycles):
        drv.Step()
    run_oracle_cycles(O, pin, cycles)
    assert "quad gather" in drv.md.lib.jb_last_transport_variant(drv.md.handle).decode()
    _compare_swarm(drv.md, O)
    _compare_fields(drv.md, O)
    assert drv.md.events == O.events


@pytest.mark.parametrize("mode", ["forced", "queues", "queues, codes gathered", "one class allowed", "no class allowed"])
@pytest.mark.parametrize("deck,overrides,cycles", [c for c in CASES if c[0].endswith("_ddmc")])
def test_cell_codes_on_all_ddmc_meshes(gpu_device, deck, overrides, cycles, mode, monkeypatch):
    """k_ddmc_all<.., cell codes> (round 6): the event loop gathers a 4-byte code per step -- the number of the
    cell's step record among the DISTINCT records of the cycle, which k_ddmc_pack counts and the kernel keeps
    in LDS -- instead of the 64-byte record.  "forced": also on the meshes small enough for the whole record
    table to sit in LDS (JB_COOP_GATHER=4).  When a mesh has more distinct records than the table holds the
    library keeps the 64-byte gather: JB_DDMC_MAX_CLASSES = 1 / 0 lower the limit so that the SMR decks (several
    classes: level x neighbour pattern) and every deck take that way out.  "queues": the library's default where
    the codes apply and at most 64 blocks are resident -- k_ddmc_q (jb_kernel_ddmc_q.hpp): finished histories
    and new photons pass through two queues in LDS, the event loop runs at full width; "forced" is k_ddmc_all on
    the same codes.  On a mesh of at most 1024 cells k_ddmc_q keeps the codes in LDS too (no vector-memory
    instruction left in its event loop); "queues, codes gathered" (JB_DDMC_LDS_CODES=0) is the form the larger
    meshes run, on every deck.  Same bits in all of them."""
    from oracle import orc
    if mode == "forced":
        monkeypatch.setenv("JB_COOP_GATHER", "4")
    elif mode.startswith("queues"):
        monkeypatch.delenv("JB_COOP_GATHER", raising=False)
        monkeypatch.setenv("JB_DDMC_QUEUES", "1")
        monkeypatch.setenv("JB_DDMC_LDS_CODES", "1" if mode == "queues" else "0")
    else:
        monkeypatch.delenv("JB_COOP_GATHER", raising=False)
        monkeypatch.setenv("JB_DDMC_MAX_CLASSES", "1" if mode == "one class allowed" else "0")
    pin = load_deck(deck, overrides)
    drv = _gpu_problem(pin, gpu_device)
    O, mesh, _ = make_oracle(load_deck(deck, overrides), orc.MATH_PORTABLE)
    for _ in range(cycles):
        drv.Step()
    run_oracle_cycles(O, pin, cycles)
    variant = drv.md.lib.jb_last_transport_variant(drv.md.handle).decode()
    if mode == "forced":
        assert "cell codes" in variant and "queues" not in variant
    elif mode.startswith("queues"):    # (the 3-D SMR deck keeps 72 blocks resident: more than the kernel's LDS table holds)
        assert "cell codes" in variant and ("queues" in variant) == (drv.md.nblocks <= 64)
        small = drv.md.nblocks <= 64 and drv.md.nblocks * drv.md.mesh.ntot <= 1024
        assert ("codes in LDS" in variant) == (mode == "queues" and small)
    elif mode == "no class allowed":
        assert "k_ddmc_all" in variant and "cell codes" not in variant
    _compare_swarm(drv.md, O)
    _compare_fields(drv.md, O)
    assert drv.md.events == O.events


def test_small_mesh_with_a_step_record_of_its_own_in_every_cell(gpu_device, monkeypatch):
    """The 1-D all-DDMC deck with a density that differs from cell to cell: 128 distinct step records on 136
    cells.  k_ddmc_q keeps the class records in LDS (<= 256) but not the codes beside them (<= 64 classes: the
    64 KB of LDS a workgroup may have), i.e. runs the gathered form on a mesh that would otherwise take the LDS
    one; same bits as the oracle."""
    from oracle import orc
    monkeypatch.delenv("JB_COOP_GATHER", raising=False)
    deck, ov = "stepdiff_ddmc", {"jaybenne/num_particles": 20000}
    pin = load_deck(deck, ov)
    drv = _gpu_problem(pin, gpu_device)
    O, mesh, _ = make_oracle(load_deck(deck, ov), orc.MATH_PORTABLE)
    rho = O.fields["rho"]
    rho *= 1.0 + 0.003 * np.arange(rho.shape[-1])[None, None, None, :]   # (denser only: every cell stays DDMC)
    O.fields["u"][...] = rho * O.fields["sie"]
    drv.md.set_field("rho", rho)
    drv.md.set_field("u", O.fields["u"])
    for _ in range(2):
        drv.Step()
    run_oracle_cycles(O, pin, 2)
    variant = drv.md.lib.jb_last_transport_variant(drv.md.handle).decode()
    nclass = drv.md.lib.jb_mesh_ddmc_classes(drv.md.handle)
    assert 64 < nclass <= 256, nclass
    assert variant == "TransportPhotons_DDMC: k_ddmc_all<1, true, cell codes, queues>" or \
        variant.endswith("k_ddmc_all<1, true, cell codes, queues>"), variant
    _compare_swarm(drv.md, O)
    _compare_fields(drv.md, O)
    assert drv.md.events == O.events


@pytest.mark.parametrize("coop", ["0", "1", "2", "4", "lds", "queues", "queues-1d", "queues-1d-gathered"])
def test_all_ddmc_photons_sitting_at_cell_faces_are_handed_to_the_general_kernel(gpu_device, coop, monkeypatch):
    """k_ddmc_all's event loop starts every step from the cell centre, which is what the albedo
    step leaves behind unless the photon sits within 2.5 eps_imc dx of a face of its cell
    (transport_utils.hpp:288-389; one sourced photon in ~1e8 does).  Those photons are listed and
    tracked by k_hybrid, launched behind k_ddmc_all on that list.  Here every tenth photon of a 3-D
    all-DDMC deck is put onto a face of its cell (lower or upper, x, y or z, inside the tolerance on
    either side of it) before the first cycle -- in the oracle's swarm too -- and all must come
    out bit-identical: accepted ones continue from the cell centre, rejected ones bounce back
    into the neighbouring cell."""
    import torch
    from oracle import orc
    if coop in ("lds", "queues-1d", "queues-1d-gathered"):   # the 1-D deck as shipped: 136 cells (k_ddmc_all: step
        # records in LDS; k_ddmc_q: the cell codes in LDS, or gathered from device memory as on a large mesh)
        deck, ov = "stepdiff_ddmc", {"jaybenne/num_particles": 20000}
        monkeypatch.delenv("JB_COOP_GATHER", raising=False)
        monkeypatch.setenv("JB_DDMC_QUEUES", "0" if coop == "lds" else "1")
        monkeypatch.setenv("JB_DDMC_LDS_CODES", "0" if coop == "queues-1d-gathered" else "1")
    else:
        if coop == "queues":            # the library's default: cell codes + the wave's photons staged through LDS queues
            monkeypatch.delenv("JB_COOP_GATHER", raising=False)
        else:
            monkeypatch.setenv("JB_COOP_GATHER", coop)
        deck, ov, _ = [c for c in CASES if c[0] == "stepdiff_ddmc" and "parthenon/mesh/nx3" in c[1]][0]
    pin = load_deck(deck, ov)
    drv = _gpu_problem(pin, gpu_device)
    O, mesh, _ = make_oracle(load_deck(deck, ov), orc.MATH_PORTABLE)
    n = O.n
    assert n == drv.md.n
    sel = np.arange(0, n, 10)
    tol = 2.5 * 1.0e6 * 2.220446049250313e-16
    moved = 0
    for k, q in enumerate(sel):
        b = int(O.sw["blk"][q])
        axis, upper = k % mesh.ndim, (k // 3) % 2
        name = "xyz"[axis]
        dx = mesh.blk_dx[b, axis]
        cell = np.floor((O.sw[name][q] - mesh.blk_xmin[b, axis]) / dx)
        face = mesh.blk_xmin[b, axis] + (cell + upper) * dx
        # inside the cell by a fraction of the tolerance (so that Xtoijk still finds this cell)
        pos = face + (-1.0 if upper else 1.0) * 0.25 * tol * dx * ((k % 4) + 0.5) / 4.0
        if np.floor((pos - mesh.blk_xmin[b, axis]) / dx) != cell:
            continue
        O.sw[name][q] = pos
        moved += 1
    assert moved > 0.9 * len(sel)
    for name in "xyz":
        drv.md.swarm[name][:n] = torch.from_numpy(O.sw[name][:n].copy()).to(drv.md.swarm[name].device)
    for _ in range(2):
        drv.Step()
    run_oracle_cycles(O, pin, 2)
    variant = drv.md.lib.jb_last_transport_variant(drv.md.handle).decode()
    assert {"lds": "k_ddmc_all<1, true, records in LDS>", "queues-1d": "k_ddmc_all<1, true, cell codes, queues, codes in LDS>",
            "queues-1d-gathered": "k_ddmc_all<1, true, cell codes, queues>",
            "queues": "k_ddmc_all<3, true, cell codes, queues"}.get(coop, "k_ddmc_all<3") in variant
    _compare_swarm(drv.md, O)
    _compare_fields(drv.md, O)
    assert drv.md.events == O.events


@pytest.mark.parametrize("deck,overrides,cycles", [CASES[0], CASES[5], CASES[8], CASES[9]])
def test_defrag_particles_restores_cell_order_and_changes_nothing_else(gpu_device, deck, overrides, cycles):
    """DefragParticles (reference jaybenne.cpp:499-509; here a counting sort of the swarm by
    block and cell, for the locality of the cell gathers) after every cycle: every photon, found by
    its creation id, ends with the same bits as in a run without it -- and as in the oracle --,
    the fields agree, and the swarm comes out ordered by (block, cell)."""
    from oracle import orc
    from jaybenne_amd import jaybenne as jb
    pin = load_deck(deck, overrides)
    drv = _gpu_problem(pin, gpu_device)
    drv.md.defrag_interval = 1
    O, mesh, _ = make_oracle(load_deck(deck, overrides), orc.MATH_PORTABLE)
    ncyc = cycles + 1
    for _ in range(ncyc):
        drv.Step()
    run_oracle_cycles(O, pin, ncyc)
    assert drv.md.defrags == ncyc
    g = drv.md.get_swarm()
    assert drv.md.n == O.n and drv.md.events == O.events
    order_g = np.argsort(g["id"], kind="stable")
    order_o = np.argsort(O.sw["id"][:O.n], kind="stable")
    assert np.array_equal(g["id"][order_g], O.sw["id"][:O.n][order_o])
    for k in g:
        a, b = g[k][order_g], O.sw[k][:O.n][order_o]
        assert np.array_equal(a, b), f"particle attribute {k} differs from the oracle after DefragParticles"
    _compare_fields(drv.md, O)
    # sorted by (block, cell of the position): the key the sort used, recomputed here
    m = drv.mesh
    b = g["blk"].astype(np.int64)
    cell = np.zeros(len(b), dtype=np.int64)
    stride = 1
    for d, name in enumerate("xyz"):
        if d < m.ndim:
            idx = np.floor((g[name] - m.blk_xmin[b, d]) * (1.0 / m.blk_dx[b, d])).astype(np.int64) + m.is_[d]
        else:
            idx = np.full(len(b), m.is_[d], dtype=np.int64)
        cell += stride * idx
        stride *= m.field_shape[3 - d]
    key = b * int(np.prod(m.field_shape[1:])) + cell
    assert np.all(np.diff(key) >= 0)
    # ... and an empty swarm is left alone
    drv.md.sv.n = 0
    assert jb.DefragParticles(drv.md) == jb.TaskStatus.complete


def test_remove_marked_particles_keeps_photons_waiting_for_their_hand_off(gpu_device):
    """Swarm::RemoveMarkedParticles (reference transport.cpp:176-178) removes what was MARKED: absorbed and
    escaped photons, and the slots whose records have been packed.  A photon that has left its rank's blocks
    but has not been packed yet (status OUTGOING / OUTGOING_ABSORBED) is still this rank's to deliver, so a
    compaction between the transport pass and the exchange -- the answer to jb_exchange's JB_ERR_CAPACITY --
    keeps it, status included, wherever it is moved."""
    from jaybenne_amd import jaybenne as jb
    drv = _gpu_problem(load_deck("stepdiff", {"jaybenne/num_particles": 5000}), gpu_device)
    drv.Step()
    md = drv.md
    n = md.n
    assert n > 1000
    rs = np.random.RandomState(11)
    status = rs.choice(np.array([0, 0, 1, 2, 3, 4], dtype=np.int32), size=n).astype(np.int32)
    status[-200:] = rs.choice(np.array([0, 3, 4], dtype=np.int32), size=200)   # movers of every live kind at the end
    import torch
    md.swarm["status"][:n].copy_(torch.from_numpy(status))
    before = md.get_swarm()
    assert jb.RemoveMarkedParticles(md) == int(np.sum((status == 0) | (status >= 3)))
    after = md.get_swarm()
    keep = (status == 0) | (status >= 3)
    order_b = np.argsort(before["id"][keep], kind="stable")
    order_a = np.argsort(after["id"], kind="stable")
    for k in before:
        assert np.array_equal(before[k][keep][order_b], after[k][order_a]), k
    assert set(np.unique(after["status"]).tolist()) == {0, 3, 4}


def test_all_ddmc_mesh_runs_the_lean_kernel(gpu_device):
    for deck, want in (("stepdiff_ddmc", "k_ddmc_all<1"), ("stepdiff_smr_hybrid", "k_hybrid<2"),
                       ("stepdiff_smr_ddmc", "k_ddmc_all<2"), ("stepdiff", "k_transport<1")):
        drv = _gpu_problem(load_deck(deck, {"jaybenne/num_particles": 3000}), gpu_device)
        drv.Step()
        assert want in drv.md.lib.jb_last_transport_variant(drv.md.handle).decode(), deck


def test_more_resident_blocks_than_the_lds_table_holds(gpu_device):
    """The DDMC kernels keep the per-block tables of up to 128 resident blocks in LDS; a mesh
    with more (here 160 blocks of 4 cells, all DDMC) runs the general kernel with the tables in
    global memory -- same bits."""
    from oracle import orc
    ov = {"parthenon/mesh/nx1": 640, "parthenon/meshblock/nx1": 4, "jaybenne/num_particles": 20000}
    pin = load_deck("stepdiff_ddmc", ov)
    drv = _gpu_problem(pin, gpu_device)
    assert drv.mesh.nblocks == 160
    O, mesh, _ = make_oracle(load_deck("stepdiff_ddmc", ov), orc.MATH_PORTABLE)
    for _ in range(2):
        drv.Step()
    run_oracle_cycles(O, pin, 2)
    assert "k_transport<1" in drv.md.lib.jb_last_transport_variant(drv.md.handle).decode()
    _compare_swarm(drv.md, O)
    _compare_fields(drv.md, O)
    assert drv.md.events == O.events


@pytest.mark.parametrize("deck,overrides", [
    ("stepdiff", {"jaybenne/num_particles": 3000}),
    ("stepdiff_smr_hybrid", {"jaybenne/num_particles": 20000}),
    ("stepdiff_ddmc", {"parthenon/mesh/nx2": 8, "parthenon/mesh/nx3": 8, "parthenon/mesh/nx1": 64,
                       "parthenon/meshblock/nx1": 32, "parthenon/meshblock/nx2": 8,
                       "parthenon/meshblock/nx3": 8, "jaybenne/num_particles": 20000})])
def test_per_event_opacity_path_bit_exact(gpu_device, deck, overrides, monkeypatch):
    """The general kernels (EOS / opacity evaluated per event from rho, sie and the photon energy,
    six separate face-probability gathers per DDMC step -- the reference's own data flow,
    transport.cpp:122-127, transport_ddmc.cpp:150-159) give the same bits as the gray fast path
    and the oracle."""
    from oracle import orc
    monkeypatch.setenv("JB_PER_EVENT_OPACITY", "1")
    pin = load_deck(deck, overrides)
    drv = _gpu_problem(pin, gpu_device)
    monkeypatch.delenv("JB_PER_EVENT_OPACITY")
    O, mesh, _ = make_oracle(load_deck(deck, overrides), orc.MATH_PORTABLE)
    drv.Step()
    run_oracle_cycles(O, pin, 1)
    _compare_swarm(drv.md, O)
    _compare_fields(drv.md, O)
    assert drv.md.events == O.events


EPBREMSS = {"mcblock/opacity_model": "ep_bremss", "mcblock/mass_scale": 1e-9,
            "mcblock/scattering_constant_value": 1e12, "jaybenne/num_particles": 4000,
            "jaybenne/do_emission": "true", "jaybenne/do_feedback": "false"}


@pytest.mark.parametrize("deck,extra", [
    ("stepdiff", {}),
    ("stepdiff_ddmc", {"mcblock/scattering_constant_value": 3e15}),
    ("stepdiff_smr", {"jaybenne/num_particles": 20000})])
def test_epbremss_opacity_bit_exact(gpu_device, deck, extra):
    """The frequency-dependent absorption model (mcblock.cpp:108-113, opacity.hpp:25): opacities
    from rho, T and the photon's energy at every event, emission from its frequency-integrated
    emissivity, non-trivial code -> CGS scales -- same bits as the oracle's statement of the same
    formulas (include/jaybenne_amd.h; singularity-opac itself is not vendored)."""
    from oracle import orc
    over = dict(EPBREMSS, **extra)
    pin = load_deck(deck, over)
    drv = _gpu_problem(pin, gpu_device)
    O, mesh, _ = make_oracle(load_deck(deck, over), orc.MATH_PORTABLE)
    # (no feedback: the material state never sees the order of the absorption atomics, so the
    # particles stay bit-identical over the cycles; compaction reorders the survivors)
    for _ in range(2):
        drv.Step()
    run_oracle_cycles(O, pin, 2)
    _compare_swarm_by_id(drv.md, O)
    _compare_fields(drv.md, O, ("tally", "fleck", "src_num", "src_ew"))
    assert drv.md.events == O.events
    assert drv.md.stats()["n_absorbed"] > 500 and O.n > 100   # absorbing, not everything absorbed


def test_model_coefficients_and_functions(gpu_device):
    """jb_initialize's EPBremss / ThomsonS coefficients and the device functions against the
    oracle's: same doubles."""
    from oracle import orc
    from jaybenne_amd import _lib
    lib = _lib.load()
    scales = dict(time_scale=3e-9, mass_scale=2e-7, length_scale=0.5, temperature_scale=1.1e3)
    p, e = _lib.Params(num_particles=10, dt=1.0), _lib.Eos(model=0, gm1=0.6, cv=1.5)
    o = _lib.Opacity(model=1, kappa=0.0, c=3e10, sb=5.67e-5, **scales)
    s = _lib.Scattering(model=1, kappa_s=0.0, apm=1.3, **scales)
    ctx = C.c_void_p()
    assert lib.jb_initialize(C.byref(p), C.byref(e), C.byref(o), C.byref(s), 0, C.byref(ctx)) == _lib.JB_COMPLETE
    out = (C.c_double * 4)()
    assert lib.jb_debug_model_coefficients(ctx, C.byref(out)) == _lib.JB_COMPLETE
    co = orc.model_coefficients(**scales)
    assert [out[0], out[1], out[2], out[3]] == [co["ep_A"], co["ep_B"], co["ep_E"], co["kappa_s_thomson"]]
    rng = np.random.default_rng(11)
    n = 20000
    x = np.stack([10 ** rng.uniform(-3, 3, n), 10 ** rng.uniform(-2, 4, n), 10 ** rng.uniform(-4, 12, n)], axis=1)
    x[:50, 1] = 0.0      # T = 0
    P = dict(opac_model=1, ep_A=co["ep_A"], ep_B=co["ep_B"], ep_E=co["ep_E"],
             kappa_s=co["kappa_s_thomson"], apm=1.3)
    orc.set_math_mode(orc.MATH_PORTABLE)
    for which in (0, 1, 2):
        got = np.empty(n)
        assert lib.jb_debug_model_eval(ctx, which, np.ascontiguousarray(x).ctypes.data, n, got.ctypes.data) == _lib.JB_COMPLETE
        want = orc.model_eval(P, which, x[:, 0], x[:, 1], x[:, 2])
        assert np.array_equal(got, want), which
    # 1 - exp(-x)
    xs = np.concatenate([10 ** rng.uniform(-300, 2, 50000), rng.uniform(0, 45, 50000), [0.0, 0.25, 40.0, 1e300]])
    got = np.empty_like(xs)
    assert lib.jb_debug_math(ctx, 11, xs.ctypes.data, xs.size, got.ctypes.data) == _lib.JB_COMPLETE
    assert np.array_equal(got, orc.math_one_minus_exp_neg(xs))
    lib.jb_finalize(ctx)


def test_absorption_emission_feedback_bit_exact(gpu_device):
    """Absorbing, emitting material with feedback (the inf.in regime): exercises absorption
    tallies, removal/compaction, the emission source and UpdateFluid."""
    from oracle import orc
    ov = {"parthenon/mesh/nx1": 16, "parthenon/meshblock/nx1": 8, "jaybenne/num_particles": 20000,
          "jaybenne/do_emission": "true", "jaybenne/do_feedback": "true",
          "mcblock/opacity_model": "constant", "mcblock/opacity_constant_value": 40.0,
          "mcblock/scattering_constant_value": 20.0, "mcblock/initial_temperature": 1.0e6}
    pin = load_deck("stepdiff", ov)
    drv = _gpu_problem(pin, gpu_device)
    O, _, _ = make_oracle(load_deck("stepdiff", ov), orc.MATH_PORTABLE)
    def compare(exact):
        _compare_swarm_by_id(drv.md, O, exact)
        sl = drv.mesh.interior()
        emitted = np.where(O.fields["src_num"][sl] > 0,
                           O.fields["src_num"][sl] * O.fields["src_ew"][sl], 0.0)
        _compare_fields(drv.md, O, ("tally", "fleck", "u"))
        _compare_fields(drv.md, O, ("edelta",), scale=emitted)

    # cycle 1: nothing depends on the order of the absorption atomics yet -> bit-exact particles
    drv.Step()
    run_oracle_cycles(O, pin, 1)
    assert drv.md.stats()["n_absorbed"] > 1000
    compare(exact=True)
    # later cycles: u (hence T, the Fleck factor and the emission weights) carries the
    # summation order of energy_delta in its last bits -> 1e-11 relative on the attributes
    dt = pin.GetReal("jaybenne", "dt")
    for cyc in (1, 2):
        drv.Step()
        O.RadiationStep(cyc * dt, dt)
        O.fields["sie"][...] = O.fields["u"] / O.fields["rho"]
    compare(exact=False)


@pytest.mark.parametrize("deck,cycles", [("inf", 12), ("inf_stiff", 10)])
def test_infinite_medium_decks_bit_exact(gpu_device, deck, cycles):
    """The reference's two equilibrium decks (inputs/inf.in: 3-D IMC, sigma_s = 1e5;
    inputs/inf_stiff.in: 1-D DDMC, sigma_a = 1e3): emission every cycle, absorption, removal.
    No feedback, so the material state never sees the order of the absorption atomics and the
    particles stay bit-identical for the whole run."""
    from oracle import orc
    pin = load_deck(deck)
    drv = _gpu_problem(pin, gpu_device)
    O, _, _ = make_oracle(load_deck(deck), orc.MATH_PORTABLE, capacity_factor=40.0)
    dt = pin.GetReal("jaybenne", "dt")
    t = 0.0
    for cyc in range(cycles):
        drv.Step()
        O.RadiationStep(t, dt)
        t += dt                      # the driver's own accumulation (mcblock_driver.cpp:46-56)
        if cyc in (0, cycles - 1):
            _compare_swarm_by_id(drv.md, O, exact=True)
    sl = drv.mesh.interior()
    emitted = O.fields["src_num"][sl] * O.fields["src_ew"][sl]
    _compare_fields(drv.md, O, ("tally", "fleck", "src_num", "src_ew"))
    _compare_fields(drv.md, O, ("edelta",), scale=emitted)
    assert drv.md.events == O.events


@pytest.mark.parametrize("deck,overrides", [
    ("stepdiff", {"parthenon/swarm/ix1_bc": "outflow", "parthenon/swarm/ox1_bc": "outflow",
                  "jaybenne/num_particles": 20000, "mcblock/scattering_constant_value": 20.0}),
    ("stepdiff_ddmc", {"parthenon/swarm/ix1_bc": "outflow", "parthenon/swarm/ox1_bc": "outflow",
                       "parthenon/swarm/ox2_bc": "outflow", "parthenon/mesh/nx2": 16,
                       "parthenon/mesh/nx1": 32, "parthenon/meshblock/nx1": 16,
                       "parthenon/meshblock/nx2": 8, "jaybenne/num_particles": 20000,
                       "mcblock/scattering_constant_value": 400.0})])
def test_outflow_boundaries_bit_exact(gpu_device, deck, overrides):
    """Photons that leave through an `outflow` swarm boundary are removed (status ESCAPED,
    compaction); the survivors equal the oracle's, 1-D IMC and 2-D DDMC."""
    from oracle import orc
    pin = load_deck(deck, overrides)
    drv = _gpu_problem(pin, gpu_device)
    O, _, _ = make_oracle(load_deck(deck, overrides), orc.MATH_PORTABLE)
    n0 = drv.md.n
    for _ in range(2):
        drv.Step()
    run_oracle_cycles(O, pin, 2)
    assert drv.md.stats()["n_escaped"] > 100 and drv.md.n < n0
    _compare_swarm_by_id(drv.md, O, exact=True)
    _compare_fields(drv.md, O, ("tally",))


def test_erf_gate_on_gpu(gpu_device):
    """The reference's own acceptance test (tst/stepdiff.py): 128 cells, 1 block, 1e5 particles,
    10 cycles, solution-weighted mean fractional error of energy_tally <= 0.05."""
    from jaybenne_amd import analysis
    pin = load_deck("stepdiff", {"parthenon/mesh/nx1": 128, "parthenon/meshblock/nx1": 128})
    drv = _gpu_problem(pin, gpu_device)
    drv.Execute()
    assert drv.ncycle == 10
    tally = np.zeros(drv.mesh.field_shape)
    tally[drv.md.gids] = drv.md.get_field("tally")
    err = analysis.analytic_errors(drv.mesh, tally, drv.time)
    assert err["mean_frac_error_weighted"] <= 0.05, err


def test_separate_tasks_match_fused_step(gpu_device):
    """TransportPhotons + CheckCompletion + EvaluateRadiationEnergy as separate tasks give the
    same tally as the fused census tally of RadiationStep."""
    from jaybenne_amd import jaybenne as jb
    pin = load_deck("stepdiff", {"jaybenne/num_particles": 5000})
    a = _gpu_problem(pin, gpu_device)
    b = _gpu_problem(load_deck("stepdiff", {"jaybenne/num_particles": 5000}), gpu_device)
    a.Step()
    dt = b.dt
    jb.UpdateDerivedTransportFields(b.md, dt)
    jb.TransportPhotons(b.md, 0.0, dt)
    assert jb.CheckCompletion(b.md, dt) == jb.TaskStatus.complete
    assert b.md.num_unfinished == 0
    assert jb.CheckCompletion(b.md, 2 * dt) == jb.TaskStatus.iterate
    assert b.md.num_unfinished == b.md.n
    jb.EvaluateRadiationEnergy(b.md)
    np.testing.assert_allclose(a.md.get_field("tally"), b.md.get_field("tally"), rtol=1e-12)
    ga, gb = a.md.get_swarm(), b.md.get_swarm()
    for k in ga:
        assert np.array_equal(ga[k], gb[k]), k


@pytest.mark.parametrize("deck,overrides,coop", [
    ("stepdiff_ddmc", {"jaybenne/num_particles": 20000}, None),        # 1-D: records in LDS
    (CASES[9][0], CASES[9][1], "1"),                                    # 3-D, quad-cooperative gather
    (CASES[9][0], CASES[9][1], "0"),                                    # 3-D, four loads per lane
    ("stepdiff_smr_hybrid", {"jaybenne/num_particles": 30000}, None),  # IMC / DDMC hybrid: k_hybrid
])
def test_separate_tasks_match_fused_step_ddmc(gpu_device, deck, overrides, coop, monkeypatch):
    """The same for TransportPhotons_DDMC: the kernels without the fused census tally
    (k_ddmc_all<.., TALLY = false, ..>, k_hybrid<.., false, ..>: what a host that schedules
    EvaluateRadiationEnergy as its own task runs, e.g. examples/handoff_mpi.cpp) leave the same
    particles, and the separate tally task the same tally, as the fused step."""
    from jaybenne_amd import jaybenne as jb
    if coop is not None:
        monkeypatch.setenv("JB_COOP_GATHER", coop)
    a = _gpu_problem(load_deck(deck, overrides), gpu_device)
    b = _gpu_problem(load_deck(deck, overrides), gpu_device)
    a.Step()
    dt = b.dt
    jb.UpdateDerivedTransportFields(b.md, dt)
    jb.TransportPhotons_DDMC(b.md, 0.0, dt)
    assert jb.CheckCompletion(b.md, dt) == jb.TaskStatus.complete
    jb.EvaluateRadiationEnergy(b.md)
    np.testing.assert_allclose(a.md.get_field("tally"), b.md.get_field("tally"), rtol=1e-12)
    ga, gb = a.md.get_swarm(), b.md.get_swarm()
    for k in ga:
        assert np.array_equal(ga[k], gb[k]), k


# ------------------------------------------------------------------------------------------------
def test_ddmc_face_probabilities_match_oracle(gpu_device):
    """UpdateDerivedTransportFields, DDMC part (jaybenne.cpp:319-489) on a 2-level mesh: interior
    faces, block faces against coarser / finer / periodic neighbours, physical boundary."""
    from jaybenne_amd import jaybenne as jb
    from oracle import orc
    for deck, ov in (("stepdiff_smr_hybrid", {"jaybenne/num_particles": 1000}),
                     ("stepdiff_ddmc", {"parthenon/mesh/nx2": 8, "parthenon/mesh/nx3": 8,
                                        "parthenon/mesh/nx1": 32, "parthenon/meshblock/nx1": 16,
                                        "parthenon/meshblock/nx2": 4, "parthenon/meshblock/nx3": 4,
                                        "jaybenne/num_particles": 1000})):
        drv = _gpu_problem(load_deck(deck, ov), gpu_device)
        O, mesh, _ = make_oracle(load_deck(deck, ov), orc.MATH_PORTABLE)
        jb.UpdateDerivedTransportFields(drv.md, drv.dt)
        O.UpdateDerivedTransportFields(drv.dt)
        m = mesh
        for d, name in enumerate(("P1", "P2", "P3")[:m.ndim]):
            sl = [slice(None)] + [slice(m.is_[dd], m.is_[dd] + m.nx[dd] + (1 if dd == d else 0))
                                  for dd in (2, 1, 0)]
            a = drv.md.get_field(name)[tuple(sl)]
            b = O.fields[name][tuple(sl)]
            assert np.array_equal(a, b), (deck, name)
            assert a.min() > 0
        assert np.array_equal(drv.md.get_field("fleck")[m.interior()], O.fields["fleck"][m.interior()])


def test_photon_reflect_bc_task(gpu_device):
    """PhotonReflectBC<BFACE> as a stand-alone task (boundaries.hpp:24-84) for all six faces."""
    import torch
    from jaybenne_amd import jaybenne as jb
    from oracle import orc
    ov = {"parthenon/mesh/nx2": 8, "parthenon/mesh/nx3": 8, "parthenon/mesh/nx1": 16,
          "parthenon/meshblock/nx1": 8, "parthenon/meshblock/nx2": 4, "parthenon/meshblock/nx3": 4,
          "jaybenne/num_particles": 3000}
    drv = _gpu_problem(load_deck("stepdiff", ov), gpu_device)
    O, mesh, _ = make_oracle(load_deck("stepdiff", ov), orc.MATH_PORTABLE)
    n = drv.md.n
    rng = np.random.default_rng(3)
    shift = rng.uniform(-0.7, 0.7, size=(3, n))       # push a good fraction beyond every face
    for q, k in enumerate(("x", "y", "z")):
        new = O.sw[k][:n] + shift[q]
        O.sw[k][:n] = new
        drv.md.swarm[k][:n] = torch.from_numpy(new).to(gpu_device)
    for face in range(6):
        jb.PhotonReflectBC(drv.md, face)
        O.PhotonReflectBC(face)
    g = drv.md.get_swarm()
    for k in ("x", "y", "z", "vx", "vy", "vz", "ip", "jp", "kp"):
        assert np.array_equal(g[k], O.sw[k][:n]), k
    assert (g["vx"] != O.sw["vx"][:n] * 0 + g["vx"]).sum() == 0


def test_c_level_radiation_step(gpu_device):
    """jb_radiation_step (the whole task list behind one C call) equals the task-by-task mirror."""
    import ctypes as C
    from jaybenne_amd import _lib
    ov = {"jaybenne/num_particles": 6000, "jaybenne/do_emission": "true",
          "mcblock/opacity_model": "constant", "mcblock/opacity_constant_value": 30.0,
          "mcblock/initial_temperature": 1.0e6, "parthenon/mesh/nx1": 16, "parthenon/meshblock/nx1": 8}
    a = _gpu_problem(load_deck("stepdiff", ov), gpu_device)
    b = _gpu_problem(load_deck("stepdiff", ov), gpu_device)
    a.Step()
    md = b.md
    md._sync_stream()
    next_id, epoch = C.c_uint64(md.next_id), C.c_uint32(md.cycle)
    _lib.check(md.lib.jb_radiation_step(md.pkg.ctx, md.handle, C.byref(md.sv), 0.0, b.dt,
                                        C.byref(next_id), C.byref(epoch), md.prefix.data_ptr()))
    assert next_id.value == a.md.next_id and epoch.value == a.md.cycle == 1
    ga, gb = a.md.get_swarm(), md.get_swarm()
    oa, ob = np.argsort(ga["id"]), np.argsort(gb["id"])
    assert a.md.n == md.n
    for k in ga:
        assert np.array_equal(ga[k][oa], gb[k][ob]), k
    sl = a.mesh.interior()
    for k in ("tally", "edelta", "u", "fleck"):   # atomic accumulation order differs run to run
        fa, fb = a.md.get_field(k)[sl], md.get_field(k)[sl]
        np.testing.assert_allclose(fa, fb, rtol=1e-12, atol=1e-12 * np.abs(fa).max(), err_msg=k)


def test_c_abi_error_paths(gpu_device):
    import ctypes as C
    from jaybenne_amd import _lib, jaybenne as jb
    drv = _gpu_problem(load_deck("stepdiff", {"jaybenne/num_particles": 1000}), gpu_device)
    md, lib = drv.md, drv.md.lib
    # particle range outside the swarm
    st = lib.jb_transport_photons(md.pkg.ctx, md.handle, C.byref(md.sv), 0.0, 1e-11, 0, md.n + 5, 0)
    assert st == _lib.JB_ERR_INVALID and b"outside the swarm" in lib.jb_last_error()
    # DDMC task without face-probability arrays
    st = lib.jb_transport_photons_ddmc(md.pkg.ctx, md.handle, C.byref(md.sv), 0.0, 1e-11, 0, md.n, 0)
    assert st == _lib.JB_ERR_INVALID
    # capacity: arrivals that do not fit
    import torch
    rec = np.zeros((md.capacity, _lib.JB_RECORD_WORDS), dtype=np.int64)
    t = torch.from_numpy(rec).to(gpu_device)
    st = lib.jb_unpack_incoming(md.pkg.ctx, md.handle, C.byref(md.sv), t.data_ptr(), md.capacity)
    assert st == _lib.JB_ERR_CAPACITY
    with pytest.raises(_lib.JaybenneError):
        _lib.check(st)
    # bad face
    assert lib.jb_photon_reflect_bc(md.pkg.ctx, md.handle, C.byref(md.sv), 7) == _lib.JB_ERR_INVALID
    # DefragParticles: null mesh, a swarm view that claims more particles than it has room for
    assert lib.jb_defrag_particles(md.pkg.ctx, None, C.byref(md.sv)) == _lib.JB_ERR_INVALID
    bad = _lib.SwarmView(n=md.capacity + 1, capacity=md.capacity)
    for name in ("x", "y", "z", "vx", "vy", "vz", "t", "w", "e", "ip", "jp", "kp", "blk", "status", "id", "rng"):
        setattr(bad, name, getattr(md.sv, name))
    assert lib.jb_defrag_particles(md.pkg.ctx, md.handle, C.byref(bad)) == _lib.JB_ERR_INVALID
    # field refresh: unknown field, sample count that is not 1 / 2 / 4 / 8, empty request
    idx = torch.zeros(4, dtype=torch.int32, device=gpu_device)
    val = torch.zeros(4, dtype=torch.float64, device=gpu_device)
    assert lib.jb_gather_cells(md.pkg.ctx, md.handle, 99, 4, idx.data_ptr(), idx.data_ptr(),
                               val.data_ptr()) == _lib.JB_ERR_INVALID
    assert lib.jb_fill_cells(md.pkg.ctx, md.handle, _lib.FIELD_IDS["u"], 1, 3, idx.data_ptr(),
                             idx.data_ptr(), idx.data_ptr(), idx.data_ptr(), None) == _lib.JB_ERR_INVALID
    assert lib.jb_fill_cells(md.pkg.ctx, md.handle, _lib.FIELD_IDS["u"], 0, 2, None, None, None,
                             None, None) == _lib.JB_COMPLETE
    # unsupported opacity model is rejected at Initialize
    p, e = _lib.Params(num_particles=10, dt=1.0), _lib.Eos(model=0, gm1=0.6, cv=1.5)
    o, s = _lib.Opacity(model=7, kappa=0.0, c=3e10, sb=5.67e-5), _lib.Scattering(model=0, kappa_s=1.0, apm=1.0)
    ctx = C.c_void_p()
    assert lib.jb_initialize(C.byref(p), C.byref(e), C.byref(o), C.byref(s), 0, C.byref(ctx)) == _lib.JB_ERR_UNSUPPORTED
    # ... and EPBremss without its code -> CGS scales
    o = _lib.Opacity(model=1, kappa=0.0, c=3e10, sb=5.67e-5)
    assert lib.jb_initialize(C.byref(p), C.byref(e), C.byref(o), C.byref(s), 0, C.byref(ctx)) == _lib.JB_ERR_INVALID
    # the energy source strategy is accepted by Initialize and rejected by SourcePhotons (sourcing.cpp:38)
    pin = load_deck("stepdiff", {"jaybenne/source_strategy": "energy", "jaybenne/num_particles": 100})
    with pytest.raises(NotImplementedError, match="Energy source strategy"):
        _gpu_problem(pin, gpu_device)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("arithmetic", [pytest.param("lean", marks=pytest.mark.lean), "exact"])
def test_full_size_invariants_c2(gpu_device, arithmetic):
    """BASELINE config C2 at full size (64 blocks of 64^3, 1e7 photons, pure IMC): too large for the
    oracle, so checked through size-independent properties -- conservation of particles and
    energy (no absorption, reflecting / periodic walls), every history ends exactly at census,
    |v| = c, tally integral = radiation energy, and bitwise run-to-run determinism of the
    particle states.  In both arithmetic variants of the kernel (the library's default and the
    exact one)."""
    import torch
    sys_path = __import__("sys").path
    root = __import__("os").path.dirname(__import__("os").path.dirname(__file__))
    if root not in sys_path:
        sys_path.insert(0, root)
    import bench
    from jaybenne_amd import mcblock

    def run():
        drv = mcblock.McblockDriver(bench.make_deck(1, 10_000_000), device=gpu_device, capacity_factor=1.2)
        assert drv.pkg.arithmetic() == arithmetic
        n0 = drv.md.n
        e0 = float(drv.md.swarm["w"][:n0].sum())
        drv.Step()
        return drv, n0, e0

    a, n0, e0 = run()
    md = a.md
    assert abs(n0 - 10_000_000) < 20_000            # stochastic rounding of 0.6 photons per cell
    assert md.n == n0
    st = md.stats()
    assert st["n_absorbed"] == st["n_escaped"] == st["n_outgoing"] == 0 and st["n_census"] == n0
    assert 1300 < st["n_events"] / n0 < 1500
    sw = md.swarm
    assert float(sw["w"][:n0].sum()) == e0
    assert bool((sw["t"][:n0] >= a.time * (1 - 1e-15)).all())
    v = torch.sqrt(sw["vx"][:n0] ** 2 + sw["vy"][:n0] ** 2 + sw["vz"][:n0] ** 2)
    assert float((v / 2.99792458e10 - 1).abs().max()) < 1e-14
    assert bool((sw["status"][:n0] == 0).all())
    sl = a.mesh.interior()
    tally = md.fields["tally"][sl]
    dv = a.mesh.cell_volume(0)
    assert float(tally.sum()) * dv == pytest.approx(e0, rel=1e-11)
    # every photon sits in the cell its indices name
    m = a.mesh
    xmin = torch.from_numpy(m.blk_xmin[md.gids]).to(gpu_device)
    dx = torch.from_numpy(m.blk_dx[md.gids]).to(gpu_device)
    blk = sw["blk"][:n0].long()
    for d, (pos, idx) in enumerate((("x", "ip"), ("y", "jp"), ("z", "kp"))):
        cell = torch.floor((sw[pos][:n0] - xmin[blk, d]) / dx[blk, d]).int() + m.ng
        assert bool((cell == sw[idx][:n0]).all())
        assert bool(((sw[idx][:n0] >= m.ng) & (sw[idx][:n0] < m.ng + m.nx[d])).all())
    # determinism: a second run from the same deck gives the same bits for every particle
    keep = {k: sw[k][:n0].clone() for k in ("x", "y", "z", "vx", "t", "rng", "blk")}
    del a
    b, n1, _ = run()
    assert n1 == n0
    for k, ref in keep.items():
        assert bool((b.md.swarm[k][:n0] == ref).all()), k


# ------------------------------------------------------------------------------------------------
def test_empty_and_ragged_inputs(gpu_device):
    """Empty swarm, zero-length ranges, fewer particles than cells, a rank-local pool that is
    mostly holes: every task must be a no-op or handle the ragged case, never fault."""
    import ctypes as C
    from jaybenne_amd import _lib, jaybenne as jb
    from oracle import orc
    # (a) no initial radiation at all: every task on an empty swarm
    ov = {"mcblock/initial_radiation": "none", "jaybenne/num_particles": 1000}
    drv = _gpu_problem(load_deck("stepdiff_ddmc", ov), gpu_device)
    md = drv.md
    assert md.n == 0
    drv.Step()
    assert md.n == 0 and md.stats()["n_events"] == 0
    assert jb.CheckCompletion(md, 1.0) == jb.TaskStatus.complete and md.num_unfinished == 0
    jb.EvaluateRadiationEnergy(md)
    assert float(md.fields["tally"].abs().max()) == 0.0
    assert jb.RemoveMarkedParticles(md) == 0
    jb.SampleDDMCBlockFace(md)
    for face in range(6):
        jb.PhotonReflectBC(md, face)
    counts = np.zeros(1, dtype=np.int64)
    _lib.check(md.lib.jb_pack_outgoing(md.pkg.ctx, md.handle, C.byref(md.sv), 0, 0, 1, None, 0,
                                       counts.ctypes.data))
    assert counts[0] == 0
    # (b) far fewer particles than cells (npc = 0.05): most cells source nothing
    ov = {"parthenon/mesh/nx2": 16, "parthenon/mesh/nx3": 16, "parthenon/mesh/nx1": 32,
          "parthenon/meshblock/nx1": 16, "parthenon/meshblock/nx2": 8, "parthenon/meshblock/nx3": 8,
          "jaybenne/num_particles": 400}
    drv = _gpu_problem(load_deck("stepdiff", ov), gpu_device)
    O, _, _ = make_oracle(load_deck("stepdiff", ov), orc.MATH_PORTABLE)
    assert 0 < drv.md.n == O.n < 8192
    drv.Step()
    run_oracle_cycles(O, load_deck("stepdiff", ov), 1)
    _compare_swarm(drv.md, O)
    # (c) a pool that is mostly holes compacts to the survivors, whatever their positions
    import torch
    n = drv.md.n
    st = torch.ones(n, dtype=torch.int32, device=gpu_device)
    keep = torch.arange(3, n, 7, device=gpu_device)
    st[keep] = 0
    ids_before = drv.md.swarm["id"][:n][keep].clone()
    drv.md.swarm["status"][:n] = st
    assert jb.RemoveMarkedParticles(drv.md) == len(keep)
    g = drv.md.get_swarm()
    assert sorted(g["id"].tolist()) == sorted(ids_before.cpu().numpy().view(np.uint64).tolist())
    assert (g["status"] == 0).all()
    # all holes
    drv.md.swarm["status"][:drv.md.n] = 1
    assert jb.RemoveMarkedParticles(drv.md) == 0


def test_defrag_policy_modes_and_results_independent_of_slot_order(gpu_device):
    """jb_defrag_policy (the default schedule of DefragParticles, include/jaybenne_amd.h): below 2^20
    photons it never sorts; DECIDE never sorts; SORT_NOW sorts (the swarm comes out ordered by block
    and cell) and counts; and a run on the default schedule ends, photon by photon (by creation id),
    with the bits of a run that never sorts -- slot order only affects speed."""
    import ctypes as C
    from jaybenne_amd import _lib
    ov = {"parthenon/mesh/nx1": 32, "parthenon/mesh/nx2": 32, "parthenon/mesh/nx3": 32,
          "parthenon/meshblock/nx1": 16, "parthenon/meshblock/nx2": 16, "parthenon/meshblock/nx3": 16,
          "jaybenne/num_particles": 1100000}
    runs = {}
    for interval in (-1, 0):
        drv = _gpu_problem(load_deck("stepdiff_ddmc", ov), gpu_device)
        drv.md.defrag_interval = interval
        for _ in range(6):
            drv.Step()
        runs[interval] = (drv.md.get_swarm(), drv.md.n, drv.md.events, drv.md.defrags)
        if interval == -1:
            md = drv.md
            md._sync_stream()
            flag = C.c_int32(7)
            # DECIDE: an answer, no sort; an unknown mode is refused
            before = md.defrags
            _lib.check(md.lib.jb_defrag_policy(md.pkg.ctx, md.handle, C.byref(md.sv), 1000, 1, C.byref(flag)))
            assert flag.value in (0, 1)
            assert md.lib.jb_defrag_policy(md.pkg.ctx, md.handle, C.byref(md.sv), 1000, 5, C.byref(flag)) == _lib.JB_ERR_INVALID
            # SORT_NOW: sorted by (block, cell)
            _lib.check(md.lib.jb_defrag_policy(md.pkg.ctx, md.handle, C.byref(md.sv), 1000, 2, C.byref(flag)))
            assert flag.value == 1 and md.defrags == before
            g = md.get_swarm()
            m = drv.mesh
            b = g["blk"].astype(np.int64)
            cell = np.zeros(len(b), dtype=np.int64)
            stride = 1
            for d, name in enumerate("xyz"):
                idx = np.floor((g[name] - m.blk_xmin[b, d]) * (1.0 / m.blk_dx[b, d])).astype(np.int64) + m.is_[d]
                cell += stride * idx
                stride *= m.field_shape[3 - d]
            assert np.all(np.diff(b * int(np.prod(m.field_shape[1:])) + cell) >= 0)
            runs[interval] = (g, md.n, md.events, md.defrags)
        del drv
    (g, n, ev, sorts), (h, m_, ev2, sorts0) = runs[-1], runs[0]
    assert n == m_ and ev == ev2 and sorts0 == 0
    og, oh = np.argsort(g["id"], kind="stable"), np.argsort(h["id"], kind="stable")
    for k in g:
        assert np.array_equal(g[k][og], h[k][oh]), k
    # a small swarm is left alone whatever its timings say
    small = _gpu_problem(load_deck("stepdiff_ddmc", {"jaybenne/num_particles": 20000}), gpu_device)
    for _ in range(8):
        small.Step()
    assert small.md.defrag_interval == -1 and small.md.defrags == 0
