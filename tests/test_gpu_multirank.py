"""The block-partitioned path with inter-rank particle hand-off, on the GPU: two ranks that share
cuda:0 and talk over gloo (one GPU box has one card; RCCL needs one device per rank, the production
launch).  Every kernel of the hand-off runs for real -- transport marks OUTGOING particles, pack,
compaction, exchange, unpack, SampleDDMCBlockFace on arrivals, completion all-reduce -- and the
union of both ranks' particles must equal the single-process CPU oracle bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from helpers import load_deck, make_oracle, run_oracle_cycles

pytestmark = pytest.mark.gpu

SMR = {"parthenon/mesh/nx1": 64, "parthenon/mesh/nx2": 32,
       "parthenon/meshblock/nx1": 16, "parthenon/meshblock/nx2": 16}
CASES = [
    ("stepdiff", {"jaybenne/num_particles": 4000}, 2),
    ("stepdiff_ddmc", {"jaybenne/num_particles": 20000, "parthenon/meshblock/nx1": 25}, 2),  # 1-D DDMC, 4 blocks
    ("stepdiff_smr_ddmc", dict(SMR, **{"jaybenne/num_particles": 30000}), 2),
    ("stepdiff_smr_hybrid", {"jaybenne/num_particles": 20000}, 1),
    ("stepdiff", {"parthenon/mesh/nx2": 8, "parthenon/mesh/nx3": 8, "parthenon/mesh/nx1": 16,
                  "parthenon/meshblock/nx1": 8, "parthenon/meshblock/nx2": 4,
                  "parthenon/meshblock/nx3": 4, "jaybenne/num_particles": 3000}, 1),
    ("stepdiff_smr_ddmc", {"parthenon/mesh/nx1": 32, "parthenon/mesh/nx2": 16, "parthenon/mesh/nx3": 16,
                           "parthenon/meshblock/nx1": 8, "parthenon/meshblock/nx2": 8,
                           "parthenon/meshblock/nx3": 8, "jaybenne/num_particles": 40000}, 2),  # 3-D SMR DDMC
    # BASELINE configs[3]: stepdiff_smr as shipped (2-D, 2 levels, 20 blocks, pure IMC)
    ("stepdiff_smr", {"jaybenne/num_particles": 20000}, 2),
    # 1-D DDMC in 32 blocks of 4 cells: a rank keeps ~10 of them resident, and particles that
    # diffuse past the halo copy in one cycle leave with a GLOBAL block id well beyond the
    # resident count (the relocation path must not index per-resident-block tables with it)
    ("stepdiff_ddmc", {"jaybenne/num_particles": 20000, "parthenon/mesh/nx1": 128,
                       "parthenon/meshblock/nx1": 4}, 2),
    # DefragParticles (every rank sorts its part of the swarm by block and cell) after every cycle:
    # arrivals are appended behind a sorted swarm, local block indices include the halo copies
    ("stepdiff_smr_hybrid", {"jaybenne/num_particles": 20000, "jaybenne/defrag_interval": 1}, 3),
    # BASELINE configs[4]'s mesh: the hybrid deck with the nested level-2 region (32 blocks, 3 levels;
    # coarse cells DDMC, both finer levels IMC), as bench.py --workload c5 builds it
    ("stepdiff_smr_hybrid", {"jaybenne/num_particles": 30000, "_level2": True}, 2),
]
LEVEL2 = ("<parthenon/static_refinement2>\nlevel = 2\nx1min = -0.125\nx1max = 0.125\n"
          "x2min = -0.125\nx2max = 0.125\nx3min = -0.25\nx3max = 0.25\n")


def _deck(case):
    deck, ov, cycles = CASES[case]
    ov = dict(ov)
    level2 = ov.pop("_level2", False)
    pin = load_deck(deck, ov)
    if level2:
        pin.load_string(LEVEL2)
    return pin, cycles



def _run_workers(procs, timeout=300):
    """Start the rank processes, wait for them and make sure none is left behind: a rank that hangs
    would keep holding the GPU (the pool allows few processes per card) and fail later tests."""
    try:
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout)
        codes = [p.exitcode for p in procs]
        assert all(c == 0 for c in codes), f"rank exit codes {codes} (None = still running after {timeout} s)"
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
                p.join(10)
            if p.is_alive():
                p.kill()
                p.join(10)

def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, outdir, decomposition="blocks", handoff="c", min_records=None):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), JB_HANDOFF=handoff)
    if min_records is not None:
        os.environ["JB_HANDOFF_MIN_RECORDS"] = str(min_records)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from jaybenne_amd import mcblock
        from jaybenne_amd.comm import Comm
        pin, cycles = _deck(case)
        drv = mcblock.McblockDriver(pin, rank=rank, nranks=world, comm=Comm(),
                                    device=torch.device("cuda", 0), capacity_factor=2.0,
                                    decomposition=decomposition)
        assert drv.decomposition == decomposition
        n0 = drv.md.n
        for _ in range(cycles):
            drv.Step()
        if min_records is not None:     # the record buffers started too small: the capacity protocol has run
            assert drv.md._chandoff.grown > 2, drv.md._chandoff.grown
        if decomposition == "blocks":   # the hand-off ran through the path asked for
            assert drv.md.handoff_path().startswith("python" if handoff == "python" else "c: jb_exchange, its two collectives as torch"), drv.md.handoff_path()
        g = drv.md.get_swarm()
        g["gblk"] = drv.md.gids[g["blk"]]
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), tally=drv.md.get_field("tally"),
                 gids=drv.md.gids, events=np.array([drv.md.events]), sourced=np.array([n0]),
                 outgoing=np.array([drv.md.stats()["n_outgoing"]]),
                 iterations=np.array([drv.md.transport_iterations_total]), **g)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", range(len(CASES)))
def test_two_ranks_equal_the_oracle(gpu_device, case, tmp_path):
    """(the hand-off through the library's one C call, jb_exchange -- the default; its two collectives are
    torch.distributed calls over gloo here: jaybenne_amd/handoff.py)"""
    _ranks_equal_the_oracle(case, 2, tmp_path)


@pytest.mark.parametrize("case,world", [(3, 2), (6, 4)])
def test_c_hand_off_grows_its_record_buffers_through_the_capacity_protocol(gpu_device, case, world, tmp_path):
    """jb_exchange with record buffers of 16 entries to start with (JB_HANDOFF_MIN_RECORDS): JB_ERR_CAPACITY comes
    out on every rank in the same call (the ranks' room travels in the count matrix), each rank grows what IT
    lacks, all call again (jaybenne_amd/handoff.py) -- and the photons are the oracle's."""
    _ranks_equal_the_oracle(case, world, tmp_path, min_records=16)


@pytest.mark.parametrize("case,world", [(0, 2), (3, 2), (7, 2), (6, 4)])
def test_ranks_equal_the_oracle_with_the_hand_off_driven_from_python(gpu_device, case, world, tmp_path):
    """The same protocol driven from Python (comm.py: count kernel, read-back, all-gather, all-to-all-v as
    separate steps; JB_HANDOFF=python), kept for A/B against the C call: same particles."""
    _ranks_equal_the_oracle(case, world, tmp_path, handoff="python")


# 2-D hybrid SMR deck (20 blocks), 3-D SMR DDMC (72 blocks), pure-IMC SMR (configs[3]: the deck
# BASELINE runs on 4 GPUs), 1-D DDMC in 32 small blocks
@pytest.mark.parametrize("case", [3, 5, 6, 7])
def test_four_ranks_equal_the_oracle(gpu_device, case, tmp_path):
    """Four ranks (still one card, gloo): every rank hands particles to several others, the
    count matrix is 4 x 4, halo copies come from up to three neighbours."""
    _ranks_equal_the_oracle(case, 4, tmp_path)


@pytest.mark.parametrize("case", [3, 6])
def test_five_ranks_equal_the_oracle(gpu_device, case, tmp_path):
    """Five ranks on the one card: this pool lets at most 6 processes share a GPU, and the test
    runner is one of them (the reference's CI runs its SMR decks on 8 MPI ranks,
    .github/workflows/ci.yml:129-140; the 8-rank hand-off logic is covered on CPU in
    test_comm_gloo.py).  20 blocks over 5 ranks: a 5 x 5 count matrix, ranks with halo copies
    from up to four neighbours."""
    _ranks_equal_the_oracle(case, 5, tmp_path)


@pytest.mark.parametrize("case,world", [(9, 4), (9, 2)])
def test_three_level_hybrid_mesh_across_ranks(gpu_device, case, world, tmp_path):
    """BASELINE configs[4]'s 3-level mesh (32 blocks) split over 4 and 2 ranks by Mesh.partition with the
    cost of mcblock.block_costs (a DDMC block costs 1 % of an IMC block): level changes, DDMC / IMC
    interfaces and rank boundaries coincide; the union of the ranks' photons equals the oracle."""
    _ranks_equal_the_oracle(case, world, tmp_path)


@pytest.mark.parametrize("case,world", [(3, 2), (9, 4), (6, 4), (8, 3), (1, 2)])
def test_replicated_mesh_split_particles_equals_the_oracle(gpu_device, case, world, tmp_path):
    """SURVEY 8e's other decomposition (jaybenne.MeshData(replicated=True)): every rank holds the whole
    mesh and sources its share of every block's photons (jb_source_photons_fill_range), one transport
    iteration per cycle, nothing handed over, the ranks' tallies summed by one all-reduce.  The
    histories are the single-process ones bit for bit (a photon's stream and every field it reads do
    not depend on which rank follows it); every rank ends with the whole tally; every rank sources
    1 / world of the photons to within one photon per block."""
    _ranks_equal_the_oracle(case, world, tmp_path, decomposition="replicated")


def _ranks_equal_the_oracle(case, world, tmp_path, decomposition="blocks", handoff="c", min_records=None):
    from oracle import orc
    sys.path.insert(0, os.path.dirname(__file__))
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, str(tmp_path), decomposition, handoff, min_records))
             for r in range(world)]
    _run_workers(procs)
    pin, cycles = _deck(case)
    O, mesh, _ = make_oracle(pin, orc.MATH_PORTABLE)
    n_sourced = O.n
    run_oracle_cycles(O, pin, cycles)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    if decomposition == "replicated":
        assert all(int(p["outgoing"][0]) == 0 for p in parts)
        assert all(int(p["iterations"][0]) == cycles for p in parts)       # one transport launch per cycle
        assert all(len(p["gids"]) == mesh.nblocks for p in parts)
        share = np.array([int(p["sourced"][0]) for p in parts])
        assert share.sum() == n_sourced and np.all(np.abs(share - n_sourced / world) <= mesh.nblocks)
    else:
        assert sum(int(p["outgoing"][0]) for p in parts) > 0, "the case must exercise the hand-off"
    ids = np.concatenate([p["id"] for p in parts])
    order = np.argsort(ids)
    oo = np.argsort(O.sw["id"][:O.n])
    assert len(ids) == O.n and np.array_equal(ids[order], O.sw["id"][:O.n][oo])
    for k in ("x", "y", "z", "vx", "vy", "vz", "t", "w", "e", "ip", "jp", "kp", "rng"):
        got = np.concatenate([p[k] for p in parts])[order]
        assert np.array_equal(got, O.sw[k][:O.n][oo]), k
    got_blk = np.concatenate([p["gblk"] for p in parts])[order]
    assert np.array_equal(got_blk, O.sw["blk"][:O.n][oo])
    sl = mesh.interior()
    for p in parts:
        a = p["tally"][sl]
        b = O.fields["tally"][p["gids"]][sl]
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=0)
    assert sum(int(p["events"][0]) for p in parts) == O.events


@pytest.mark.lean
@pytest.mark.parametrize("case", [4, 6])
def test_lean_arithmetic_does_not_depend_on_the_partition(gpu_device, case, tmp_path):
    """The library's default arithmetic (lean) across ranks: the union of two ranks' particles
    equals the single-process run of the same library bit for bit -- the operations a history
    sees do not depend on which rank tracks it -- and both are within the stated tolerance of the
    oracle (tests/test_gpu_lean.py)."""
    from oracle import orc
    from jaybenne_amd import mcblock
    from test_gpu_lean import _compare_within_tolerance
    sys.path.insert(0, os.path.dirname(__file__))
    pin, cycles = _deck(case)
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case, str(tmp_path))) for r in range(2)]
    _run_workers(procs)
    drv = mcblock.McblockDriver(pin, device=gpu_device)
    assert drv.pkg.arithmetic() == "lean"
    for _ in range(cycles):
        drv.Step()
    one = drv.md.get_swarm()
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(2)]
    assert sum(int(p["outgoing"][0]) for p in parts) > 0
    ids = np.concatenate([p["id"] for p in parts])
    order, o1 = np.argsort(ids), np.argsort(one["id"])
    assert np.array_equal(ids[order], one["id"][o1])
    for k in ("x", "y", "z", "vx", "vy", "vz", "t", "w", "e", "ip", "jp", "kp", "rng"):
        assert np.array_equal(np.concatenate([p[k] for p in parts])[order], one[k][o1]), k
    pin, _ = _deck(case)
    O, mesh, _ = make_oracle(pin, orc.MATH_PORTABLE)
    run_oracle_cycles(O, pin, cycles)
    _compare_within_tolerance(one, O.sw, O.n, mesh, pin.GetReal("jaybenne", "dt"), by_id=True)


FEEDBACK = dict(SMR, **{"jaybenne/num_particles": 30000, "jaybenne/do_emission": "true",
                        "jaybenne/do_feedback": "true", "mcblock/opacity_model": "constant",
                        "mcblock/opacity_constant_value": 20.0})


def _feedback_worker(rank, world, port, outdir, decomposition="blocks", overrides=None):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from jaybenne_amd import mcblock
        from jaybenne_amd.comm import Comm
        drv = mcblock.McblockDriver(load_deck("stepdiff_smr_hybrid", dict(FEEDBACK, **(overrides or {}))), rank=rank,
                                    nranks=world, comm=Comm(), device=torch.device("cuda", 0),
                                    capacity_factor=8.0, decomposition=decomposition)
        if decomposition == "blocks":
            assert len(drv.md.resident_gids) > drv.md.nowned        # halo copies are in use
        for _ in range(3):
            drv.Step()
        g = drv.md.get_swarm()
        g["gblk"] = drv.md.gids[g["blk"]]
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), gids=drv.md.gids,
                 u=drv.md.fields["u"].cpu().numpy(), resident=drv.md.resident_gids,
                 tally=drv.md.get_field("tally"), fleck=drv.md.get_field("fleck"),
                 edelta=drv.md.get_field("edelta"), **g)
    finally:
        dist.destroy_process_group()


def test_two_ranks_with_material_feedback(gpu_device, tmp_path):
    """Absorbing / emitting hybrid IMC-DDMC problem on a two-level mesh with the material energy
    fed back every cycle: internal_energy is refreshed in the ghost zones and in the halo copies
    through jb_gather_cells / all-to-all / jb_fill_cells.  Cycle 1 is independent of the order
    of the absorption atomics; afterwards u carries it in its last bits (1e-11 on attributes)."""
    from oracle import orc
    sys.path.insert(0, os.path.dirname(__file__))
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_feedback_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    _run_workers(procs)
    pin = load_deck("stepdiff_smr_hybrid", FEEDBACK)
    O, mesh, _ = make_oracle(pin, orc.MATH_PORTABLE, capacity_factor=8.0)
    # the emission count per cell scales with 1 / (blocks in the calling rank's MeshData)
    # (sourcing.cpp:68-69): 20 blocks over 2 ranks
    O.emission_blocks_in_call = mesh.nblocks // 2
    run_oracle_cycles(O, pin, 3)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(2)]
    assert all(len(p["gids"]) == mesh.nblocks // 2 for p in parts)
    ids = np.concatenate([p["id"] for p in parts])
    order = np.argsort(ids)
    oo = np.argsort(O.sw["id"][:O.n])
    assert len(ids) == O.n and np.array_equal(ids[order], O.sw["id"][:O.n][oo])
    for k in ("ip", "jp", "kp", "rng"):
        assert np.array_equal(np.concatenate([p[k] for p in parts])[order], O.sw[k][:O.n][oo]), k
    for k in ("x", "y", "z", "vx", "vy", "vz", "t", "w", "e"):
        got = np.concatenate([p[k] for p in parts])[order]
        np.testing.assert_allclose(got, O.sw[k][:O.n][oo], rtol=1e-11, atol=0, err_msg=k)
    for p in parts:
        # owned blocks and halo copies alike, ghost zones included
        np.testing.assert_allclose(p["u"], O.fields["u"][p["resident"]], rtol=1e-12, atol=0)
        sl = mesh.interior()
        np.testing.assert_allclose(p["fleck"][sl], O.fields["fleck"][p["gids"]][sl], rtol=1e-12)
        np.testing.assert_allclose(p["tally"][sl], O.fields["tally"][p["gids"]][sl], rtol=1e-11)


def test_replicated_mesh_with_material_feedback(gpu_device, tmp_path):
    """The same absorbing / emitting problem with the mesh replicated on three ranks: every rank sources
    its share of every cell's emission photons, rank 0 alone books the emitted energy, absorption adds
    to each rank's own energy_delta, ONE all-reduce sums them, and UpdateFluid then forms the same
    internal energy on every rank (the ranks' u arrays are equal bit for bit -- the all-reduce gives
    every rank the same sum; against the oracle 1e-12: the order of the sum)."""
    from oracle import orc
    sys.path.insert(0, os.path.dirname(__file__))
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_feedback_worker, args=(r, 3, port, str(tmp_path), "replicated")) for r in range(3)]
    _run_workers(procs)
    pin = load_deck("stepdiff_smr_hybrid", FEEDBACK)
    O, mesh, _ = make_oracle(pin, orc.MATH_PORTABLE, capacity_factor=8.0)
    O.emission_blocks_in_call = mesh.nblocks          # (sourcing.cpp:68-69: the whole mesh in one MeshData)
    run_oracle_cycles(O, pin, 3)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(3)]
    ids = np.concatenate([p["id"] for p in parts])
    order = np.argsort(ids)
    oo = np.argsort(O.sw["id"][:O.n])
    assert len(ids) == O.n and np.array_equal(ids[order], O.sw["id"][:O.n][oo])
    for k in ("ip", "jp", "kp", "rng"):
        assert np.array_equal(np.concatenate([p[k] for p in parts])[order], O.sw[k][:O.n][oo]), k
    for k in ("x", "y", "z", "vx", "vy", "vz", "t", "w", "e"):
        got = np.concatenate([p[k] for p in parts])[order]
        np.testing.assert_allclose(got, O.sw[k][:O.n][oo], rtol=1e-11, atol=0, err_msg=k)
    sl = mesh.interior()
    for p in parts:
        assert np.array_equal(p["u"], parts[0]["u"])
        np.testing.assert_allclose(p["u"], O.fields["u"], rtol=1e-12, atol=0)
        np.testing.assert_allclose(p["tally"][sl], O.fields["tally"][sl], rtol=1e-11)


def test_replicated_mesh_feedback_without_the_emission_source(gpu_device, tmp_path):
    """do_emission = false, do_feedback = true on two ranks with the mesh replicated, three cycles: nothing
    resets energy_delta then (sourcing.cpp:41-43 returns before :165-166), it ACCUMULATES over the cycles
    and UpdateFluid deposits the running sum every cycle (jaybenne.cpp:583-615) -- so behind cycle 1's
    all-reduce every rank holds the global sum, and cycle 2's all-reduce must add only the cycle's own
    absorptions to it, not nranks copies of what is already there.  energy_delta and u of every
    rank against the single-process oracle."""
    from oracle import orc
    sys.path.insert(0, os.path.dirname(__file__))
    ov = {"jaybenne/do_emission": "false"}
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_feedback_worker, args=(r, 2, port, str(tmp_path), "replicated", ov)) for r in range(2)]
    _run_workers(procs)
    pin = load_deck("stepdiff_smr_hybrid", dict(FEEDBACK, **ov))
    O, mesh, _ = make_oracle(pin, orc.MATH_PORTABLE, capacity_factor=8.0)
    run_oracle_cycles(O, pin, 3)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(2)]
    sl = mesh.interior()
    assert np.abs(O.fields["edelta"][sl]).max() > 0.0, "the case must absorb something"
    for p in parts:
        np.testing.assert_allclose(p["edelta"][sl], O.fields["edelta"][sl], rtol=1e-11, atol=0)
        np.testing.assert_allclose(p["u"], O.fields["u"], rtol=1e-11, atol=0)
        assert np.array_equal(p["u"], parts[0]["u"])
    ids = np.concatenate([p["id"] for p in parts])
    order = np.argsort(ids)
    oo = np.argsort(O.sw["id"][:O.n])
    assert len(ids) == O.n and np.array_equal(ids[order], O.sw["id"][:O.n][oo])
    for k in ("ip", "jp", "kp", "rng"):
        assert np.array_equal(np.concatenate([p[k] for p in parts])[order], O.sw[k][:O.n][oo]), k


def _read_photon_dumps(prefix, nranks):
    rec = np.dtype([("id", "<u8"), ("x", "<f8"), ("vx", "<f8"), ("t", "<f8"), ("w", "<f8"), ("rng", "<u8"),
                    ("gblk", "<i4"), ("ip", "<i4")])
    parts = []
    for r in range(nranks):
        with open(f"{prefix}.{r}.bin", "rb") as fh:
            n = int(np.frombuffer(fh.read(8), dtype="<u8")[0])
            parts.append(np.frombuffer(fh.read(), dtype=rec, count=n))
    out = np.concatenate(parts)
    return out[np.argsort(out["id"])]


def test_c_level_mpi_handoff(gpu_device, tmp_path):
    """examples/handoff_mpi.cpp: the hand-off driven from C++ over MPI (no Python, no PyTorch in
    the rank processes), three ranks.  With halo copies planned by the C++ mirror of the task
    interface (include/jaybenne_amd.hpp: PlanHalo, PlanHaloRefresh -- what the Parthenon adapter's
    mesh-view builder uses too) a cycle takes TWO transport iterations and every photon comes out
    bit-identical to the single-process oracle; without them (halo_rings = 0: a rank keeps only
    the blocks it owns) every rank-boundary crossing goes through jb_pack_outgoing ->
    MPI_Alltoallv -> jb_unpack_incoming, dozens of iterations per cycle, the same photons.  Photon
    count, total weight, census time and the tally integral are checked by the program every
    cycle."""
    import re
    import shutil
    import subprocess
    from oracle import orc
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mpiexec = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"
    if not os.path.exists(mpiexec) or not os.path.exists("/opt/conda/lib/libmpi.so"):
        pytest.skip("no MPI installation in this image")
    build = subprocess.run(["make", "-C", os.path.join(root, "examples"), "mpi"], capture_output=True, text=True)
    assert build.returncode == 0, build.stdout + build.stderr
    ov = {"parthenon/mesh/nx1": 128, "parthenon/meshblock/nx1": 16, "jaybenne/num_particles": 200000}
    O, mesh, _ = make_oracle(load_deck("stepdiff", ov), orc.MATH_PORTABLE)
    run_oracle_cycles(O, load_deck("stepdiff", ov), 3)
    order = np.argsort(O.sw["id"][:O.n])
    per_cycle = {}
    for rings in (1, 0):
        prefix = str(tmp_path / f"photons{rings}")
        run = subprocess.run([mpiexec, "-n", "3", os.path.join(root, "examples", "handoff_mpi"), "16", "8",
                              "200000", "3", str(rings), prefix], capture_output=True, text=True, timeout=240)
        assert run.returncode == 0 and "HANDOFF OK" in run.stdout, run.stdout + run.stderr
        assert run.stdout.count(" ok") == 3
        per_cycle[rings] = float(re.search(r"\(([0-9.]+) per cycle\)", run.stdout).group(1))
        g = _read_photon_dumps(prefix, 3)
        assert len(g) == O.n
        assert np.array_equal(g["id"], O.sw["id"][:O.n][order])
        for k in ("x", "vx", "t", "w", "rng", "ip"):
            assert np.array_equal(g[k], O.sw[k][:O.n][order]), (rings, k)
        assert np.array_equal(g["gblk"], O.sw["blk"][:O.n][order])
    assert per_cycle[1] == 2.0            # transport, hand-off of the photons that ended in a halo copy, done
    assert per_cycle[0] > 20.0            # one iteration per rank-boundary crossing of the most persistent photon
