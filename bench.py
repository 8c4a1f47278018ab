#!/usr/bin/env python3
"""Headline benchmark: particle-histories/s of one radiation cycle on `stepdiff`
(BASELINE.json metric), on N GPUs of one node.

Workload (BASELINE.json configs[1], SURVEY.md section 8d "C2"): inputs/stepdiff.in as a pure-IMC
uniform 3-D mesh, 64 meshblocks of 64^3 cells (256^3 cells) and 1e7 particles PER GPU; for N > 1
the block grid grows to (8,4,4), (8,8,4), (8,8,8) blocks at fixed cell size, blocks are dealt to
ranks in Z-order (one octant each at N = 8) and particles that cross into another rank's blocks
are handed over through RCCL -- weak scaling.  A step = one RadiationStep (reference
jaybenne.cpp:68-151): derived fields, transport of every photon to census incl. hand-off and the
completion test, census tally.  Inputs are resident in HBM before the timed region.

`python bench.py --gpus N` works as typed: for N > 1 without a launcher environment it starts
N fresh rank processes itself (python -m torch.distributed.run, rendezvous on 127.0.0.1) BEFORE
anything in this process touches the GPU, relays rank 0's line and exits with the children's
code.  Under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it is a rank.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_DDMC_STEP = 64.0   # one cell record {f sigma_a, sigma, six leak opacities} per DDMC step
FP64_VALU_PEAK_TF = 78.6     # vector FP64, 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz
BYTES_PER_HISTORY = 168.0    # 84 B read + 68 B write-back + 16 B census tally RMW
BYTES_PER_EVENT_IMC = 24.0   # rho, sie, fleck gathered per event (not LDS-staged)
FLOPS_PER_EVENT = 200.0      # FP64 flop-equivalents per IMC event


def block_grid(ngpus: int):
    """Blocks of the weak-scaled headline mesh.  The domain keeps its four block columns in x (the axis
    of the stepdiff temperature step and of the reflecting walls) and grows in the periodic directions y
    and z: every rank's contiguous Z-order range is then a 4 x 4 x 4 cube of blocks with 32 hot and 32
    cold ones -- the one-GPU problem, coupled to its neighbours through the particle hand-off.  (Every
    block starts with the same number of photons whatever its temperature -- the `uniform` source
    strategy, reference sourcing.cpp:68-69 -- so the ranks' photon counts are equal either way; growing
    in x would give every other rank a problem without the step in it.)"""
    return {1: (4, 4, 4), 2: (4, 8, 4), 4: (4, 8, 8), 8: (4, 16, 8)}.get(ngpus) or (4, 4 * ngpus, 4)


def make_deck(ngpus: int, particles_per_gpu: int, block_nx: int = 64, workload: str = "c2"):
    """c2: stepdiff.in pure IMC, 64 x 64^3 blocks per GPU (the headline workload).
    c3: stepdiff_ddmc.in, 3-D 128^3 cells in 8 x 64^3 blocks (sigma dx = 7.8: every step DDMC).
    c3-1d: stepdiff_ddmc.in as shipped but 128 cells in one block (tally-contention stress)."""
    from jaybenne_amd.deck import load_deck
    if workload == "c1":     # BASELINE configs[0]: the reference's own regression case (tst/stepdiff.py)
        return load_deck("stepdiff", {"jaybenne/num_particles": particles_per_gpu * ngpus,
                                      "parthenon/mesh/nx1": 128, "parthenon/meshblock/nx1": 128})
    if workload == "c3-1d":
        return load_deck("stepdiff_ddmc", {"jaybenne/num_particles": particles_per_gpu * ngpus,
                                           "parthenon/mesh/nx1": 128, "parthenon/meshblock/nx1": 128})
    if workload == "c3":
        ov = {"jaybenne/num_particles": particles_per_gpu * ngpus}
        # (development knob, tools/dev: JB_BENCH_C3_MESH="nx,block_nx,ndim" measures other table sizes)
        nx, bnx, nd = (int(v) for v in os.environ.get("JB_BENCH_C3_MESH", "128,64,3").split(","))
        for d in range(nd):
            ov[f"parthenon/mesh/nx{d + 1}"] = nx
            ov[f"parthenon/meshblock/nx{d + 1}"] = bnx
        return load_deck("stepdiff_ddmc", ov)
    if workload == "c4":     # stepdiff_smr.in as shipped: 2-D, 20 blocks of 32^2, 2 levels, pure IMC
        ov = {"jaybenne/num_particles": particles_per_gpu * ngpus}
        # (development knob, tools/dev: JB_BENCH_C4_MESH="nx1,nx2,nx3,block_nx" runs the deck in 3-D)
        if os.environ.get("JB_BENCH_C4_MESH"):
            n1, n2, n3, bnx = (int(v) for v in os.environ["JB_BENCH_C4_MESH"].split(","))
            for d, nd in enumerate((n1, n2, n3)):
                ov[f"parthenon/mesh/nx{d + 1}"] = nd
                ov[f"parthenon/meshblock/nx{d + 1}"] = bnx
        return load_deck("stepdiff_smr", ov)
    if workload == "c5":     # stepdiff_smr_hybrid.in + a nested level-2 region
        ov = {"jaybenne/num_particles": particles_per_gpu * ngpus}
        # (development knob, tools/dev: JB_BENCH_C5_MESH="nx1,nx2,nx3,block_nx" runs the deck in 3-D)
        if os.environ.get("JB_BENCH_C5_MESH"):
            n1, n2, n3, bnx = (int(v) for v in os.environ["JB_BENCH_C5_MESH"].split(","))
            for d, nd in enumerate((n1, n2, n3)):
                ov[f"parthenon/mesh/nx{d + 1}"] = nd
                ov[f"parthenon/meshblock/nx{d + 1}"] = bnx
        pin = load_deck("stepdiff_smr_hybrid", ov)
        pin.load_string("<parthenon/static_refinement2>\nlevel = 2\nx1min = -0.125\nx1max = 0.125\n"
                        "x2min = -0.125\nx2max = 0.125\nx3min = -0.25\nx3max = 0.25\n")
        return pin
    nb = block_grid(ngpus)
    ov = {"jaybenne/num_particles": particles_per_gpu * ngpus}
    for d in range(3):
        ext = nb[d] / 4.0                      # dx = 1/256 for every N (with 64^3 blocks)
        ov[f"parthenon/mesh/nx{d + 1}"] = nb[d] * block_nx
        ov[f"parthenon/meshblock/nx{d + 1}"] = block_nx
        ov[f"parthenon/mesh/x{d + 1}min"] = -0.5 * ext
        ov[f"parthenon/mesh/x{d + 1}max"] = 0.5 * ext
    return load_deck("stepdiff", ov)


def cpu_baseline(sample_particles: int, block_nx: int):
    """The oracle (CPU port of the reference algorithm, libm arithmetic) on the same workload with
    a bounded number of particles, on this box's host cores."""
    from oracle import orc
    from oracle.harness import make_oracle
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = min(avail, 16)   # the GPU box grants a 16-core share per GPU
    pin = make_deck(1, sample_particles, block_nx)
    O, _, _ = make_oracle(pin, orc.MATH_LIBM, threads=threads)
    dt = pin.GetReal("jaybenne", "dt")
    n0 = O.n
    t0 = time.perf_counter()
    O.RadiationStep(0.0, dt)
    wall = time.perf_counter() - t0
    out = {"value": n0 / wall, "unit": "particle-histories/s", "cores": threads, "kind": "port",
           "sample": f"1 cycle of the same mesh with {n0} particles (OpenMP over particles, "
                     f"{O.events / wall:.3e} events/s, {wall:.1f} s)"}
    # BASELINE configs[0] (the reference's own CPU-runnable case: stepdiff, 1-D,
    # 128 cells in one block, 1e5 particles) on ONE thread -- the analogue of mcblock with one MPI
    # rank on Kokkos Serial, the reference's default build
    from jaybenne_amd.deck import load_deck
    pin1 = load_deck("stepdiff", {"parthenon/mesh/nx1": 128, "parthenon/meshblock/nx1": 128})
    O1, _, _ = make_oracle(pin1, orc.MATH_LIBM, threads=1)
    n1 = O1.n
    t0 = time.perf_counter()
    O1.RadiationStep(0.0, pin1.GetReal("jaybenne", "dt"))
    w1 = time.perf_counter() - t0
    out["serial_c1"] = {"value": n1 / w1, "unit": "particle-histories/s", "cores": 1,
                        "sample": f"configs[0]: 1 cycle, {n1} particles, 1 thread "
                                  f"({O1.events / w1:.3e} events/s, {w1:.1f} s)"}
    return out


def accuracy(device, threads: int):
    """north_star's accuracy clause on BASELINE configs[0] (the reference's own regression case,
    tst/stepdiff.py: 1-D, 128 cells, 1e5 photons, 10 cycles): the HIP path and the CPU restatement
    with the reference's libm arithmetic run the same streams; both are scored with the
    reference's metric (tst/regression_test.py:383-406) against the analytic erf profile, and
    against each other in units of the Monte Carlo noise of a cell."""
    from jaybenne_amd import analysis, mcblock
    from jaybenne_amd.deck import load_deck
    from oracle import orc
    from oracle.harness import make_oracle, run_oracle_cycles
    ov = {"parthenon/mesh/nx1": 128, "parthenon/meshblock/nx1": 128}
    drv = mcblock.McblockDriver(load_deck("stepdiff", ov), device=device)
    arithmetic = drv.pkg.arithmetic()
    drv.Execute()
    sl = drv.mesh.interior()
    g = drv.md.get_field("tally")
    e_gpu = analysis.analytic_errors(drv.mesh, g, drv.time)["mean_frac_error_weighted"]
    pin = load_deck("stepdiff", ov)
    O, mesh, _ = make_oracle(pin, orc.MATH_LIBM, threads=threads)
    t_end = run_oracle_cycles(O, pin, drv.ncycle)
    c = O.fields["tally"]
    e_cpu = analysis.analytic_errors(mesh, c, t_end)["mean_frac_error_weighted"]
    # per-cell noise: n census photons of equal weight w in a cell -> sigma = w sqrt(n) / dV
    w = float(O.sw["w"][:O.n].max())
    dv = float(mesh.cell_volume(0))
    sigma = np.sqrt(np.maximum(c[sl] * dv / w, 1.0)) * w / dv
    z = np.abs(g[sl] - c[sl]) / sigma
    return {"config": "BASELINE configs[0]: stepdiff 1-D, 128 cells, 1e5 photons, 10 cycles",
            "metric": "weighted mean fractional error vs the analytic erf profile "
                      "(tst/regression_test.py:383-406), gate 0.05",
            "arithmetic": arithmetic,
            "gpu_error": e_gpu, "cpu_libm_error": e_cpu, "gate": 0.05,
            "gpu_minus_cpu": e_gpu - e_cpu,
            "max_cell_difference_in_sigma": float(z.max()),
            "rms_cell_difference_in_sigma": float(np.sqrt((z * z).mean())),
            "note": "same random streams; the HIP path evaluates log / sincos by table and (lean "
                    "arithmetic, the default) face distances / position updates within 4e-15 (relative) of "
                    "the CPU path's libm / IEEE operations: a history parts ways with its twin "
                    "only where a last-bit difference flips a branch, and the tally moves only if "
                    "that photon ends the cycle in another cell.  Stated tolerance "
                    "(tests/test_gpu_accuracy.py): gpu_error <= cpu_libm_error + 0.01 and every "
                    "cell within 6 sigma of its Monte Carlo noise; measured values above."}


def lean_vs_exact_after_10_cycles(device):
    """What the default (lean) arithmetic guarantees over MANY cycles, measured in this run: the two
    variants of the gray IMC kernel follow the same 1e5 photons (3-D, 32^3 cells in 8 blocks: the deck of
    tests/test_gpu_lean.py::test_lean_against_exact_over_ten_cycles) through the ten cycles the reference's
    own test runs (tst/stepdiff.py:29-55)."""
    from jaybenne_amd import mcblock
    from jaybenne_amd.deck import load_deck
    ov = {"parthenon/mesh/nx1": 32, "parthenon/mesh/nx2": 32, "parthenon/mesh/nx3": 32,
          "parthenon/meshblock/nx1": 16, "parthenon/meshblock/nx2": 16, "parthenon/meshblock/nx3": 16,
          "jaybenne/num_particles": 100000}
    res = {}
    for mode in ("lean", "exact"):
        drv = mcblock.McblockDriver(load_deck("stepdiff", ov), device=device)
        drv.pkg.set_arithmetic(mode)
        xs = []
        for _ in range(10):
            drv.Step()
            xs.append(drv.md.swarm["x"][:drv.md.n].cpu().numpy().copy())
        g = drv.md.get_swarm()
        res[mode] = (g, xs, drv.md.get_field("tally")[drv.mesh.interior()], drv.mesh)
        del drv
    (g, xl, tl, mesh), (h, xe, te, _) = res["lean"], res["exact"]
    same = g["rng"] == h["rng"]              # same stream state = same sequence of events so far
    size = float(np.max(np.asarray(mesh.gmax) - np.asarray(mesh.gmin)))
    drift = [float(np.abs(a - b)[same].max() / size) for a, b in zip(xl, xe)]
    return {"config": "stepdiff 3-D, 32^3 cells in 8 blocks, 1e5 photons, 10 cycles, lean vs exact arithmetic "
                      "on the same streams",
            "max_position_difference_over_domain_by_cycle": [float(f"{d:.3e}") for d in drift],
            "histories_resequenced": int((~same).sum()), "histories": int(len(same)),
            "photons_ending_in_another_cell": int(((g["ip"] != h["ip"]) | (g["jp"] != h["jp"]) | (g["kp"] != h["kp"])
                                                   | (g["blk"] != h["blk"])).sum()),
            "tally_cells_differing_by_more_than_1e-9": int((np.abs(tl - te) > 1e-9 * np.abs(te).max()).sum()),
            "tally_cells": int(tl.size),
            "note": "1e-9 per cycle is the stated tolerance (one cycle; 1e-8 after two); beyond that two "
                    "roundings of one history separate like nearby trajectories of any chaotic system -- "
                    "about one decade per cycle, the same rate at which the CPU path's libm and portable "
                    "flavours separate; a history is re-sequenced only where a difference flips a comparison"}


LOG_DIR = os.path.join(ROOT, "gpurun_out", "bench_logs")


def live_counters(args):
    """Counters of the tracking kernel(s) measured NOW: four child runs of this command for ONE step under
    `rocprofv3 --kernel-trace --pmc <group>` (separate passes, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
    WRITE_SIZE do not fit one pass, an SQ group is eight counters), the counters summed over the tracking kernel's
    launches of that step.  HBM bytes: KB x 1024, raw (the guide's x2 correction of FETCH_SIZE on gfx950 is
    calibrated for 16 B / lane coalesced streams, not for this kernel's 8-byte gathers and stores).  Returns None
    when the profiler is not there, fails or takes too long -- the line then quotes the committed summary, labelled."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof) or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None          # (no profiler, or this run is itself being profiled)
    groups = {"fetch": "FETCH_SIZE GRBM_GUI_ACTIVE", "write": "WRITE_SIZE",
              "sq": "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY",
              "f64": "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64"}
    kernels = ("k_transport", "k_imc_cell", "k_ddmc_all", "k_ddmc_q", "k_hybrid")
    c, ms = {}, {}
    tmp = tempfile.mkdtemp(prefix="jb_counters_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", JB_BENCH_CHILD="1")
    try:
        for name, counters in groups.items():
            d = os.path.join(tmp, name)
            cmd = [prof, "--kernel-trace", "--pmc"] + counters.split() + ["--output-format", "csv", "-d", d, "-o", "run",
                   "--", sys.executable, os.path.abspath(__file__), "--workload", args.workload,
                   "--particles-per-gpu", str(args.particles_per_gpu), "--block-nx", str(args.block_nx),
                   "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-other-variant", "--no-live-counters"]
            if args.arithmetic:
                cmd += ["--arithmetic", args.arithmetic]
            res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=90)
            if res.returncode != 0:
                return None
            seen = False
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if any(k in r["Kernel_Name"] for k in kernels):
                        c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                        seen = True
            if not seen:
                return None
            dur = 0.0
            for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if any(k in r["Kernel_Name"] for k in kernels):
                        dur += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
            ms[name] = dur
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    try:
        simd_cycles = 1024.0 * (c["GRBM_GUI_ACTIVE"] / 8.0) * (ms["sq"] / ms["fetch"])
        lane_util = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
        fr = {"valu_issue_frac": min(1.0, 4.0 * c["SQ_ACTIVE_INST_VALU"] / simd_cycles),
              "fp64_counter_frac": ((2.0 * c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"])
                                    * 64.0 * lane_util / (ms["f64"] * 1e-3) / (FP64_VALU_PEAK_TF * 1e12)),
              "valu_lane_utilisation": lane_util,
              "wave_time_fraction_waiting_on_memory": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
              "effective_clock_GHz": c["GRBM_GUI_ACTIVE"] / 8.0 / (ms["fetch"] * 1e-3) / 1e9,
              "kernel_ms_under_rocprof": ms["sq"]}
    except (KeyError, ZeroDivisionError):
        fr = {}
    return {"fetch_bytes": c["FETCH_SIZE"] * 1024.0, "write_bytes": c["WRITE_SIZE"] * 1024.0,
            "bytes": (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0, "fractions": fr, "raw": c,
            "how": "four child runs of this command for one step (first cycle) under rocprofv3 --kernel-trace --pmc "
                   "<FETCH_SIZE GRBM_GUI_ACTIVE | WRITE_SIZE | SQ group | FP64 group>, summed over the tracking "
                   "kernel's launches; HBM bytes = KB x 1024, raw"}


def self_launch(args) -> int:
    """--gpus N > 1 from a plain `python bench.py`: N fresh rank processes, started before this
    process has imported torch or made any HIP call (re-executing a process that has initialised
    the GPU is not allowed on this pool).  The ranks supervise their workers exactly as they do
    under the driver's own `python -m torch.distributed.run` (supervise())."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    return proc.returncode if (proc.returncode != 0 or line is not None) else 1


def _tail(path: str, n: int = 25) -> str:
    try:
        with open(path, "r", errors="replace") as fh:
            return "".join(fh.readlines()[-n:])
    except OSError as e:
        return f"({e})\n"


def supervise(args) -> int:
    """A rank of an N > 1 run (under `python -m torch.distributed.run`): it never touches the GPU
    itself.  It starts a WORKER process that does (first over RCCL), watches it, and agrees with the
    other ranks -- over a CPU-only gloo group on the launcher's rendezvous -- on the outcome:

      * every worker returns 0: rank 0 relays the worker's JSON line;
      * a worker exits non-zero, is killed, or the attempt runs out of time: its rank posts an
        abort key in the rendezvous store, every rank kills its worker within a second, rank 0
        prints the tail of every rank's log, and -- unless JB_BENCH_BACKEND pinned the backend --
        FRESH workers repeat the run over gloo (records through host memory), the line labelled
        `"backend": "gloo (rccl failed: ...)"`;
      * both attempts fail: exit code 1, diagnostics on stderr.  Nothing waits longer than
        JB_BENCH_BUDGET_S (default 540 s) in total, whatever a collective does.
    """
    import subprocess
    from datetime import timedelta
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    t_start = time.time()
    budget = float(os.environ.get("JB_BENCH_BUDGET_S", "540"))
    dist.init_process_group("gloo", timeout=timedelta(seconds=60))   # CPU only: no HIP call
    from torch.distributed.distributed_c10d import _get_default_store
    store = _get_default_store()
    os.makedirs(LOG_DIR, exist_ok=True)
    pinned = os.environ.get("JB_BENCH_BACKEND")
    attempts = [pinned] if pinned else ["nccl", "gloo"]
    failure_note, rc_final, line = None, 1, None
    for ai, backend in enumerate(attempts):
        # time for this attempt: RCCL gets at most 55 % of what is left when a fallback follows it
        left = budget - (time.time() - t_start)
        limit = left * (0.55 if ai + 1 < len(attempts) else 1.0) - 10.0
        if limit < 20.0:
            failure_note = (failure_note or "") + f"; no time left for attempt {ai} ({backend})"
            break
        port_t = torch.zeros(1, dtype=torch.int64)
        if rank == 0:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port_t[0] = sk.getsockname()[1]
        dist.broadcast(port_t, 0)
        env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
        env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(int(port_t[0])), JB_BENCH_BACKEND=backend,
                   JB_BENCH_WORKER="1", JB_BENCH_ATTEMPT=str(ai))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        log_path = os.path.join(LOG_DIR, f"rank{rank}.attempt{ai}.{backend}.log")
        out_path = os.path.join(LOG_DIR, f"rank{rank}.attempt{ai}.{backend}.out")
        key = f"jb_abort_{ai}"

        def post_abort(msg: str) -> None:
            # The FIRST failure names the attempt's reason (compare-and-set on an empty key): a victim --
            # a worker whose collective broke because a peer died -- must not overwrite the culprit.
            # Every rank's own reason is kept beside it, with its time, and rank 0 prints them all.
            store.set(f"{key}_r{rank}", f"{time.time():.3f} rank {rank}: {msg}")
            store.compare_set(key, "", f"rank {rank}: {msg}")
        with open(log_path, "w") as log, open(out_path, "w") as out:
            log.write(f"[supervisor] rank {rank}/{world} attempt {ai} backend {backend} limit {limit:.0f} s\n")
            log.flush()
            child = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                     env=env, stdout=out, stderr=log)
            t0, rc, why = time.time(), None, ""
            while rc is None:
                time.sleep(0.05)     # (short: the order in which the supervisors SEE their workers die is
                rc = child.poll()    # what tells the rank that failed from the ranks it took down)
                if rc is not None:
                    break
                aborted = store.check([key])
                timed_out = time.time() - t0 > limit
                if aborted or timed_out:
                    why = "another rank failed" if aborted else f"no result after {limit:.0f} s"
                    if timed_out and not aborted:
                        post_abort(why)
                    child.terminate()
                    try:
                        child.wait(timeout=5)
                    except subprocess.TimeoutExpired:
                        child.kill()
                        child.wait()
                    rc = child.returncode if child.returncode not in (0, None) else -15
                    log.write(f"[supervisor] worker stopped: {why}\n")
            if rc != 0 and not why:
                post_abort(f"worker exit code {rc}")
                log.write(f"[supervisor] worker exit code {rc}\n")
        bad = torch.tensor([1 if rc != 0 else 0], dtype=torch.int64)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad[0]) == 0:
            rc_final = 0
            if rank == 0:
                for ln in open(out_path, "r", errors="replace"):
                    if ln.startswith("{") and '"metric"' in ln:
                        line = ln.strip()
                if line is None:
                    rc_final = 1
                    print(f"bench: attempt {ai} ({backend}) ended without a result line", file=sys.stderr)
            break
        # every rank's own reason with the time its supervisor saw it; the EARLIEST is the attempt's reason
        # (identical on all ranks: the keys are complete once the all-reduce above has returned)
        every = []
        for r in range(world):
            if store.check([f"{key}_r{r}"]):
                every.append(store.get(f"{key}_r{r}").decode())
        every.sort(key=lambda ln: float(ln.split(" ", 1)[0]))
        try:
            reason = every[0].split(" ", 1)[1] if every else store.get(key).decode()
        except Exception:
            reason = "unknown"
        failure_note = ((failure_note + "; ") if failure_note else "") + f"{backend}: {reason}"
        if rank == 0:
            print(f"bench: attempt {ai} over {backend} failed ({reason}); per-rank logs in {LOG_DIR}:",
                  file=sys.stderr)
            for ln in every:
                print(f"bench:   {ln}", file=sys.stderr)
            for r in range(world):
                print(f"---- rank {r} ----\n" + _tail(os.path.join(LOG_DIR, f"rank{r}.attempt{ai}.{backend}.log")),
                      file=sys.stderr)
    ok = torch.tensor([rc_final], dtype=torch.int64)
    dist.broadcast(ok, 0)
    rc_final = int(ok[0])
    if rank == 0 and line is not None and rc_final == 0:
        d = json.loads(line)
        d["ranks"] = world
        if failure_note:
            d["backend"] = f"gloo ({failure_note.replace('nccl', 'rccl failed')})"
        print(json.dumps(d), flush=True)
    dist.destroy_process_group()
    return rc_final


def fake_worker(args) -> int:
    """JB_BENCH_FAKE_WORKER=1 (tests/test_bench_supervisor.py, CPU only): the worker's process-group
    life cycle without a GPU -- rendezvous over gloo, a barrier per 'step', a result line -- with the
    failures the supervisor has to survive injected through the environment."""
    from datetime import timedelta
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("JB_BENCH_BACKEND", "nccl")
    print(f"[fake worker] rank {rank}/{world} backend {backend}", file=sys.stderr, flush=True)
    if backend == "nccl" and os.environ.get("JB_BENCH_FAKE_NCCL_FAILS") == "1":
        print("[fake worker] RCCL refused the uneven all_to_all_single (injected)", file=sys.stderr, flush=True)
        return 3
    dist.init_process_group("gloo", timeout=timedelta(seconds=90))
    kill = os.environ.get("JB_BENCH_FAKE_KILL", "")       # "rank@step[@backend]"
    hang = os.environ.get("JB_BENCH_FAKE_HANG", "")       # "rank@step"
    for step in range(args.warmup + args.steps):
        if kill:
            kr, ks, *kb = kill.split("@")
            if int(kr) == rank and int(ks) == step and (not kb or kb[0] == backend):
                print(f"[fake worker] rank {rank} dies at step {step} (injected)", file=sys.stderr, flush=True)
                os._exit(17)
        if hang and int(hang.split("@")[0]) == rank and int(hang.split("@")[1]) == step:
            print(f"[fake worker] rank {rank} hangs at step {step} (injected)", file=sys.stderr, flush=True)
            time.sleep(3600)
        dist.barrier()
        time.sleep(0.05)
    if rank == 0:
        print(json.dumps({"metric": "particle-histories/s (whole node) on stepdiff", "value": 1.0,
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "config": {"parallelism": f"fake worker over {backend}"}}), flush=True)
    dist.destroy_process_group()
    return 0


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--particles-per-gpu", type=int, default=10_000_000,
                    help="10000000 = BASELINE configs[1] per GPU (default); 12500000 = north_star's "
                         "target invocation (1e8 particles on 8 GPUs)")
    ap.add_argument("--block-nx", type=int, default=64)
    ap.add_argument("--cpu-sample", type=int, default=2_500_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="c2", choices=["c1", "c2", "c3", "c3-1d", "c4", "c5"],
                    help="c2 = headline (BASELINE configs[1]); c3* = DDMC side measurements")
    ap.add_argument("--no-accuracy", action="store_true")
    ap.add_argument("--force-exchange", action="store_true",
                    help="1 GPU: run the hand-off phase as well, in a one-rank group (nothing moves; "
                         "shows the fixed cost of the exchange per transport iteration)")
    ap.add_argument("--defrag-interval", type=int, default=-1,
                    help="DefragParticles (sort of the swarm by cell): -1 (default) on the library's "
                         "schedule (jb_defrag_policy), k > 0 after every k-th cycle, 0 never; matters "
                         "for long runs (profiles/r04_long_run.json), not for the few cycles timed here")
    ap.add_argument("--decomposition", default="auto", choices=("auto", "blocks", "replicated"),
                    help="several GPUs: 'blocks' = meshblocks dealt to ranks by estimated tracking cost, "
                         "photons handed over through RCCL (the reference's decomposition); 'replicated' = "
                         "every rank holds the whole mesh and follows its share of every block's photons, "
                         "one all-reduce of the tally per cycle; 'auto' (default) = replicated only where "
                         "the mesh is tiny and no contiguous split of its blocks balances (configs[4])")
    ap.add_argument("--handoff", default=os.environ.get("JB_HANDOFF", "c"), choices=("c", "c-torch", "c-rccl", "python"),
                    help="several GPUs, block partition: how photons are handed to other ranks.  'c' (default) = the "
                         "library's one C call jb_exchange per transport iteration, over an RCCL communicator of its "
                         "own (falls back, labelled, to the process group's collectives through callbacks: "
                         "'c-torch'); 'python' = the same protocol driven step by step from Python (comm.py)")
    ap.add_argument("--no-blocks-variant", action="store_true",
                    help="several GPUs, c5: skip the extra run on the block partition when 'auto' chose the "
                         "replicated mesh (config.blocks_variant)")
    ap.add_argument("--no-live-counters", action="store_true",
                    help="do not measure roofline.traffic / valu_issue_frac / fp64_counter_frac with four child runs "
                         "under rocprofv3 (1 GPU, default line); the line then quotes the committed counter summary, "
                         "labelled")
    ap.add_argument("--no-other-variant", action="store_true",
                    help="skip the one extra step in the other arithmetic variant (profiling runs)")
    ap.add_argument("--arithmetic", choices=("lean", "exact"), default=None,
                    help="arithmetic of the gray IMC tracking step (default: the library's, lean)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(self_launch(args))
    if args.gpus > 1 and os.environ.get("JB_BENCH_WORKER") != "1":
        sys.exit(supervise(args))          # a rank of the launcher: it watches a worker (see there)
    if os.environ.get("JB_BENCH_FAKE_WORKER") == "1":
        sys.exit(fake_worker(args))

    t_import = time.perf_counter()
    import torch
    import torch.distributed as dist
    from jaybenne_amd import mcblock

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with "
                         "python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    # JB_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
    # (ranks then share devices and records travel through host memory); production is RCCL.
    backend = os.environ.get("JB_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(torch.cuda.device_count(), 1) if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    comm = None
    if world == 1 and args.force_exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
        else:
            dist.init_process_group(backend, rank=0, world_size=1)
        from jaybenne_amd.comm import Comm
        comm = Comm(device=device)
    def rank_log(msg: str) -> None:
        # (stderr of a supervised worker is gpurun_out/bench_logs/rank<k>.attempt<i>.<backend>.log)
        if world > 1:
            print(f"[rank {rank} +{time.perf_counter() - t_import:7.1f} s] {msg}", file=sys.stderr, flush=True)

    if world > 1:
        from datetime import timedelta
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a collective that does not complete in 90 s raises (and takes the worker down) instead of
        # sitting out torch's default 10 minutes: the supervisor then falls back or reports
        rank_log(f"device {dev_index} = {torch.cuda.get_device_name(dev_index)}, backend {backend}: rendezvous")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device, timeout=timedelta(seconds=90))
        else:
            dist.init_process_group(backend, timeout=timedelta(seconds=90))
        from jaybenne_amd.comm import Comm
        comm = Comm(device=device)
        comm.barrier()
        rank_log("process group up, first barrier passed")

    pin = make_deck(args.gpus, args.particles_per_gpu, args.block_nx, args.workload)
    drv = mcblock.McblockDriver(pin, rank=rank, nranks=world, comm=comm, device=device,
                                capacity_factor=1.5 if world == 1 else 3.0,
                                decomposition=args.decomposition if world > 1 else "blocks")
    md = drv.md
    rank_log(f"mesh: {md.mesh.nblocks} blocks in all, {md.nowned} owned + {md.nblocks - md.nowned} halo copies "
             f"resident ({drv.decomposition}); {md.n} photons sourced here")
    # photons and estimated tracking work per rank at the start (what the decomposition is judged by)
    cost = mcblock.block_costs(md.mesh, pin, drv.mcb)
    photons_by_rank = np.zeros(world, dtype=np.int64)
    photons_by_rank[rank] = md.n
    if comm is not None and world > 1:
        photons_by_rank = comm.allreduce_sum_int64(photons_by_rank)
    if drv.decomposition == "replicated":
        work_by_rank = np.full(world, float(cost.sum()) / world)
    else:
        owner = md.mesh.owner if world > 1 else np.zeros(md.mesh.nblocks, dtype=np.int32)
        work_by_rank = np.bincount(owner, weights=cost, minlength=world)
    md.force_exchange = bool(args.force_exchange)
    md.defrag_interval = int(args.defrag_interval)
    md.handoff = args.handoff
    if comm is not None and world > 1 and drv.decomposition != "replicated":
        # (the hand-off's transport -- the library's own RCCL communicator -- is made here, before any cycle, not
        # inside the first exchange of the first cycle: a driver that asks for no warm-up still times no bootstrap)
        md.ensure_handoff()
        rank_log("hand-off: " + md.handoff_path())

    def sync_all():
        torch.cuda.synchronize(device)
        if comm is not None:
            comm.barrier()
            torch.cuda.synchronize(device)

    if args.arithmetic is not None:
        drv.pkg.set_arithmetic(args.arithmetic)
    kill = os.environ.get("JB_BENCH_KILL_RANK", "")   # "rank@step": the failure rehearsal of tools/dev
    for w in range(args.warmup):
        if kill and int(kill.split("@")[0]) == rank and int(kill.split("@")[1]) == w:
            rank_log("dies here (JB_BENCH_KILL_RANK)")
            os._exit(17)
        drv.Step()
        if w == 0:
            rank_log(f"first cycle done: {md.transport_iterations_total} transport iterations, "
                     f"{md.handoff_records} records handed over by this rank, {md.n} photons here now")
    if os.environ.get("JB_PHASE_TIMES"):     # diagnostic: per-phase wall time (adds syncs)
        md.phase_times = {}
    sync_all()
    ev0 = md.events
    md.kernel_events = []
    md.handoff_records, md.exchange_seconds, md.transport_iterations_total = 0, 0.0, 0
    md.transport_wait_seconds, md.collective_seconds = 0.0, 0.0
    histories = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        histories += md.n          # photons alive at the start of the cycle, this rank
        drv.Step()
    sync_all()
    wall = time.perf_counter() - t0
    events = md.events - ev0
    if comm is not None:
        wall = comm.allreduce_max_float(wall)
        histories, events = (int(v) for v in comm.allreduce_sum_int64(np.array([histories, events])))

    # the other arithmetic variant of the gray IMC kernel, one step, for the record (1 GPU only;
    # outside the timed region above)
    other = None
    main_variant = md.lib.jb_last_transport_variant(md.handle).decode()
    main_stats = md.stats()
    main_stats_launches = int(md.transport_iterations_total) + args.warmup * max(1, int(md.transport_iterations_total) // max(args.steps, 1))
    # (read before the extra step below moves them)
    iters = int(md.transport_iterations_total)
    handoff_records = int(md.handoff_records)
    exchange_s, wait_s = float(md.exchange_seconds), float(md.transport_wait_seconds)
    coll_s = float(md.collective_seconds)
    if args.gpus == 1 and not md.pkg.Param("use_ddmc") and not args.no_other_variant:
        kept = list(md.kernel_events)
        mode = md.pkg.arithmetic()
        md.pkg.set_arithmetic("exact" if mode == "lean" else "lean")
        n_before = md.n
        sync_all()
        t1 = time.perf_counter()
        drv.Step()
        sync_all()
        other = {"arithmetic": md.pkg.arithmetic(), "ms_per_step": 1e3 * (time.perf_counter() - t1),
                 "value": n_before / (time.perf_counter() - t1), "steps": 1,
                 "kernel": md.lib.jb_last_transport_variant(md.handle).decode()}
        md.pkg.set_arithmetic(mode)
        md.kernel_events = kept

    handoff_path = md.handoff_path() if (comm is not None and drv.decomposition != "replicated") else "none (nothing is handed over)"

    # BASELINE configs[4] on several GPUs: 'auto' answers with the replicated mesh (no contiguous split of its 32
    # blocks balances better than 1.23 x the mean) -- SURVEY 8e calls that a cross-check, north_star mandates the
    # block partition with hand-off: run that too, a few steps, and report it beside
    blocks_variant = None
    if (world > 1 and args.workload == "c5" and args.decomposition == "auto" and drv.decomposition == "replicated"
            and not args.no_blocks_variant):
        rank_log("the same workload on the block partition (config.blocks_variant)")
        drv_b = mcblock.McblockDriver(make_deck(args.gpus, args.particles_per_gpu, args.block_nx, args.workload),
                                      rank=rank, nranks=world, comm=comm, device=device, capacity_factor=3.0,
                                      decomposition="blocks")
        mb = drv_b.md
        mb.defrag_interval = int(args.defrag_interval)
        mb.handoff = args.handoff
        drv_b.Step()                       # warm-up
        sync_all()
        mb.handoff_records, mb.transport_iterations_total = 0, 0
        k_b = max(1, min(args.steps, 5))
        hist_b = 0
        tb = time.perf_counter()
        for _ in range(k_b):
            hist_b += mb.n
            drv_b.Step()
        sync_all()
        wall_b = comm.allreduce_max_float(time.perf_counter() - tb)
        hist_b, rec_b = (int(v) for v in comm.allreduce_sum_int64(np.array([hist_b, mb.handoff_records])))
        owner_b = mb.mesh.owner
        work_b = np.bincount(owner_b, weights=cost, minlength=world)
        blocks_variant = {"decomposition": "blocks", "value": hist_b / wall_b, "ms_per_step": 1e3 * wall_b / k_b,
                          "steps": k_b, "warmup": 1,
                          "estimated_work_per_rank_max_over_mean": float(work_b.max() / work_b.mean()),
                          "transport_iterations_per_step": mb.transport_iterations_total / k_b,
                          "handoff_records_per_step": rec_b / k_b, "handoff_path": mb.handoff_path()}
        mb.close()
        del drv_b, mb
        torch.cuda.empty_cache()

    # hand-off statistics of the timed steps (all ranks): records are 104 bytes
    if comm is not None:
        handoff_records = int(comm.allreduce_sum_int64(np.array([handoff_records]))[0])
        exchange_s = comm.allreduce_max_float(exchange_s)
        wait_s = comm.allreduce_max_float(wait_s)
        coll_s = comm.allreduce_max_float(coll_s)

    if rank == 0:
        # dominant kernel: k_transport, timed with HIP events on its stream (rank 0's launches)
        kt = [(a.elapsed_time(b) * 1e-3, n) for a, b, n in md.kernel_events]
        k_time = sum(t for t, _ in kt)
        k_hist = sum(n for t, n in kt)
        ev_per_hist = events / max(histories, 1)
        k_events = k_hist * ev_per_hist
        ddmc_bound = args.workload in ("c3", "c3-1d")            # the two regimes
        per_event = 72.0 if args.workload in ("c3", "c3-1d", "c5") else BYTES_PER_EVENT_IMC
        k_bytes = k_hist * BYTES_PER_HISTORY + k_events * per_event
        variant = main_variant
        variant = ("TransportPhotons_DDMC: " if md.pkg.Param("use_ddmc") else "TransportPhotons: ") + variant
        # counters of this very command under rocprofv3 (separate --pmc passes), if a committed
        # summary matches workload and size: labelled as read from that file, not measured now
        pmc, pmc_file = None, None
        for rnd in ("r06", "r06_exact", "r05", "r05_exact", "r04", "r04_exact", "r03", "r03_exact", "r02", "r02_exact", "r01_g"):
            f = os.path.join(ROOT, "profiles", f"{rnd.split('_exact')[0]}_pmc_summary_{args.workload}"
                                               f"{'_exact' if rnd.endswith('_exact') else ''}.json")
            try:
                tr = json.load(open(f))
                # (kernel names of the summaries carry NDIM, DDMC, TALLY, GRAY, EXACT[, LEAN]: five
                # arguments = a kernel from before the lean variant existed, i.e. exact arithmetic)
                kargs = tr.get("kernel", "").split("<")[-1].rstrip("> ").split(",")
                tr_lean = len(kargs) == 6 and kargs[-1].strip() == "true"
                tr_lean = tr_lean or "k_imc_cell" in tr.get("kernel", "") or "cell-local" in tr.get("kernel", "")
                same_arith = tr_lean == variant.endswith(("true>", "lean>")) or "k_ddmc_all" in tr.get("kernel", "")
                if (tr["workload"] == args.workload and tr["particles_per_gpu"] == args.particles_per_gpu
                        and args.block_nx == 64 and args.gpus == 1 and same_arith):
                    pmc, pmc_file = tr, os.path.relpath(f, ROOT)
                    break
            except (OSError, KeyError, ValueError):
                continue
        fp64 = k_events * FLOPS_PER_EVENT / k_time / 1e12 if k_time > 0 else 0.0
        l2_gbs = k_bytes / k_time / 1e9 if k_time > 0 else 0.0
        if ddmc_bound:
            # DDMC regime: the bound reported is HBM -- what the kernel HAS to move per launch is
            # the particle stream (168 B per history) and its per-cell table once (cell codes: 4 B per cell
            # incl. ghosts; the 64-byte record forms: 64 B) -- and `frac` is that rate over 8 TB/s, small by
            # the nature of the path (a history is ~34 dependent steps of ~110 FP64-heavy instructions each for
            # 168 bytes).  What the steps gather through L1 / L2 is reported beside it, as a rate, against the
            # guide's figure for a table of that size -- a secondary figure, not the bound.
            codes = "cell codes" in variant
            in_lds = "records in LDS" in variant
            cells_all = float(md.nblocks) * float(md.mesh.field_shape[1] * md.mesh.field_shape[2] * md.mesh.field_shape[3])
            per_cell = 4.0 if codes else 64.0
            alg = k_hist * BYTES_PER_HISTORY + len(kt) * cells_all * per_cell
            hbm = alg / k_time / 1e9 if k_time > 0 else 0.0
            per_step = 0.0 if in_lds else (4.0 if codes else BYTES_PER_DDMC_STEP)
            gather = k_events * per_step / k_time / 1e9 if k_time > 0 else 0.0
            table_mb = cells_all * per_cell / 1e6
            roof = {"bound": "hbm", "achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": hbm / HBM_PEAK_GBS,
                    "traffic": pmc["hbm_bytes_per_launch"] if pmc else None,
                    "definition": "(168 B per history: 84 B read, 68 B write-back, 16 B tally + the kernel's "
                                  f"per-cell table read once: {per_cell:.0f} B per cell) / k_ddmc_all time (HIP events, "
                                  "this run), against 8 TB/s",
                    "algorithmic_hbm_bytes_per_launch": alg / max(len(kt), 1),
                    "gathered_GBps": gather,
                    "gathered_table_MB": table_mb,
                    "gathered_definition": ("nothing: the step records of all cells sit in LDS" if in_lds else
                                            f"DDMC steps x {per_step:.0f} B "
                                            + ("(one 4-byte cell code per step; the distinct step records, "
                                               f"{int(md.lib.jb_mesh_ddmc_classes(md.handle))} of them, are read from LDS)" if codes
                                               else "(one 64-byte step record per step)")
                                            + " / kernel time; the guide's gather rates for comparison "
                                              "(MI355X_MICROARCH.md, 'Indexed rows'): 16.8-18.8 TB/s for rows "
                                              "shared through L2, 8.6 TB/s from a 38 MB table, 7.4-7.9 TB/s from "
                                              "a 151 MB one"),
                    "lds_served_GBps": k_events * 64.0 / k_time / 1e9 if (k_time > 0 and (codes or in_lds)) else 0.0}
        else:
            cells_all = float(md.nblocks) * float(md.mesh.field_shape[1] * md.mesh.field_shape[2] * md.mesh.field_shape[3])
            roof = {"bound": "fp64_valu", "achieved": fp64, "peak": FP64_VALU_PEAK_TF,
                    "unit": "TFLOP/s", "frac": fp64 / FP64_VALU_PEAK_TF,
                    "traffic": pmc["hbm_bytes_per_launch"] if pmc else None,
                    # what HBM has to move per launch: the particle stream + the per-cell mean free paths once
                    "algorithmic_hbm_bytes_per_launch": (k_hist * BYTES_PER_HISTORY + len(kt) * cells_all * 16.0) / max(len(kt), 1),
                    "definition": "events x 200 FP64 flop-equivalents (SURVEY 8d) / k_transport time "
                                  "(HIP events, this run), against the vector FP64 peak; the IMC "
                                  "regime is bound by VALU instruction issue, not by HBM",
                    "l2_served_GBps": l2_gbs,
                    "l2_served_definition": "168 B per history + 24 B of cell gathers per event / "
                                            "kernel time: almost all served by L2, NOT an HBM rate"}
        if pmc and "valu_issue_frac" in pmc:
            # the bounds the counters themselves support (the 200-flop figure above is the
            # survey's convention): read from the committed summary of this command, not measured now
            roof["valu_issue_frac"] = pmc["valu_issue_frac"]
            roof["fp64_counter_frac"] = pmc["fp64_counter_frac"]
            roof["valu_issue_definition"] = ("share of the SIMDs' cycles spent issuing VALU instructions "
                                             "(SQ_ACTIVE_INST_VALU x waves per SIMD / SQ_WAVE_CYCLES): how close the "
                                             "kernel is to the issue limit for the instruction stream it executes; "
                                             "fp64_counter_frac = FP64 flop the counters saw (fma 2, add / mul 1, x lanes "
                                             "in use) / launch time / 78.6 TF/s; both from " + pmc_file)
        roof.update({"kernel": variant, "kernel_ms_avg": 1e3 * k_time / max(len(kt), 1),
                     "launches": len(kt), "events_per_launch": k_events / max(len(kt), 1),
                     "algorithmic_bytes_per_history": BYTES_PER_HISTORY + per_event * ev_per_hist})
        # what the kernel's own counters say about THIS run (no profiler): lanes in use per 64-lane pass of
        # the event loop, passes and service phases, and the bytes its lanes stored (13 attribute stores per
        # finished history + the hand-off records); the --pmc summaries below are the cross-check
        n_launch = max(len(kt), 1)
        finished = sum(main_stats[k] for k in ("n_census", "n_absorbed", "n_escaped", "n_outgoing"))
        roof["measured_in_run"] = {
            "lanes_per_wave_pass": main_stats["n_events"] / max(main_stats["n_wave_passes"], 1),
            "event_loop_lane_utilisation": main_stats["n_events"] / max(64 * main_stats["n_wave_passes"], 1),
            "wave_passes_per_launch": main_stats["n_wave_passes"] / max(main_stats_launches, 1),
            "service_phases_per_launch": main_stats["n_wave_services"] / max(main_stats_launches, 1),
            "stored_bytes_per_launch": 84.0 * finished / max(main_stats_launches, 1),
            "note": "cumulative counters of the library since the process started (warm-up and timed launches "
                    "alike) divided by the number of launches they cover"}
        roof["traffic_source"] = ((pmc_file + " (rocprofv3 --pmc passes of this command; not measured in this run)")
                                  if pmc else None)
        if args.gpus == 1 and not args.no_live_counters and not args.no_cpu_baseline:
            # (the default run: measure them now -- the GPU is idle, the timed region is over)
            live = live_counters(args)
            if live is not None:
                roof["traffic"] = live["bytes"]
                roof["traffic_source"] = "measured in this run: " + live["how"]
                roof["traffic_fetch_write"] = [live["fetch_bytes"], live["write_bytes"]]
                if live["fractions"]:
                    roof.update({k: v for k, v in live["fractions"].items()})
                    roof["valu_issue_definition"] = ("measured in this run (the child passes of traffic_source): "
                                                     "valu_issue_frac = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), "
                                                     "clamped to 1; fp64_counter_frac = FP64 flop the counters saw (fma 2, add / mul 1, "
                                                     "x lanes in use) / launch time / 78.6 TF/s")
                roof["counters_measured_in_run"] = live["raw"]
        if roof.get("traffic") and roof.get("algorithmic_hbm_bytes_per_launch"):
            roof["wasted_traffic_ratio"] = roof["traffic"] / roof["algorithmic_hbm_bytes_per_launch"]
        if pmc:
            roof["counters"] = {"source": pmc_file + " (rocprofv3 --pmc passes of this command; "
                                                     "not measured in this run)",
                                **{k: pmc[k] for k in pmc if k not in ("workload", "particles_per_gpu",
                                                                       "counters", "command", "hbm_note")}}
        out = {
            "metric": "particle-histories/s (whole node) on stepdiff",
            "value": histories / wall,
            "unit": "particle-histories/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * wall / args.steps,
            "value_per_gpu": histories / wall / args.gpus,
            "backend": "rccl" if (backend == "nccl" and world > 1) else (backend if world > 1 else "none (one rank)"),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": {
                "c2": "stepdiff pure-IMC, uniform 3-D mesh, "
                      f"{md.mesh.nblocks} meshblocks of {args.block_nx}^3 cells, "
                      f"{args.particles_per_gpu * args.gpus:.3g} particles, 1 cycle per step "
                      "(BASELINE.json configs[1] per GPU)",
                "c1": "[c1] stepdiff as the reference's test runs it (1-D, 128 cells, 1 block), "
                      f"{args.particles_per_gpu * args.gpus:.3g} particles (BASELINE.json configs[0])",
                "c3": "[c3] stepdiff_ddmc (all-DDMC, tau_ddmc = 5), uniform 3-D mesh, "
                      f"{md.mesh.nblocks} meshblocks of {args.block_nx}^3 cells, "
                      f"{args.particles_per_gpu * args.gpus:.3g} particles, 1 cycle per step "
                      "(BASELINE.json configs[2], SURVEY 8d C3b)",
                "c3-1d": "[c3-1d] stepdiff_ddmc deck geometry (1-D, 128 cells, all-DDMC), "
                         f"{args.particles_per_gpu * args.gpus:.3g} particles, 1 cycle per step "
                         "(SURVEY 8d C3a)",
                "c4": "[c4] stepdiff_smr as shipped (2-D, 2 levels, pure IMC), "
                      f"{md.mesh.nblocks} meshblocks, {args.particles_per_gpu * args.gpus:.3g} particles "
                      "(BASELINE.json configs[3])",
                "c5": "[c5] stepdiff_smr_hybrid + nested level-2 region (2-D, 3 levels, IMC/DDMC hybrid), "
                      f"{md.mesh.nblocks} meshblocks, {args.particles_per_gpu * args.gpus:.3g} particles "
                      "(BASELINE.json configs[4])"}[args.workload],
                       "blocks_per_gpu": md.nowned, "halo_blocks_per_gpu": md.nblocks - md.nowned,
                       "particles_per_gpu": args.particles_per_gpu,
                       "defrag_interval": int(args.defrag_interval),
                       "defrag_sorts_in_run": int(md.defrags),
                       "kernel_ms_by_step": [round(1e3 * t, 2) for t, _ in kt][:64],
                       "decomposition": drv.decomposition if world > 1 else "one rank",
                       "blocks_variant": blocks_variant,
                       "photons_per_rank_min": int(photons_by_rank.min()),
                       "photons_per_rank_max": int(photons_by_rank.max()),
                       "estimated_work_per_rank_max_over_mean": float(work_by_rank.max() / work_by_rank.mean()),
                       "parallelism": (f"whole mesh on each of {args.gpus} rank(s), every block's photons dealt "
                                       f"to the ranks by stream id, one {'RCCL' if backend == 'nccl' else backend} "
                                       "all-reduce of energy_tally / energy_delta per cycle"
                                       if (world > 1 and drv.decomposition == "replicated") else
                                       f"meshblocks over {args.gpus} rank(s), "
                                       f"{'RCCL' if backend == 'nccl' else backend} particle hand-off")},
            "events_per_s": events / wall,
            "events_per_history": ev_per_hist,
            "transport_iterations_per_step": iters / max(args.steps, 1),
            "handoff": {"path": handoff_path,
                        "records_per_step": handoff_records / max(args.steps, 1),
                        "bytes_per_step": 104.0 * handoff_records / max(args.steps, 1),
                        "exchange_ms_per_step_max_rank": 1e3 * exchange_s / max(args.steps, 1),
                        "transport_wait_ms_per_step_max_rank": 1e3 * wait_s / max(args.steps, 1),
                        "collectives_ms_per_step_max_rank": 1e3 * coll_s / max(args.steps, 1),
                        "note": "particles handed to another rank (all ranks summed); exchange = "
                                "count kernel + read-back + count all-gather + pack + all-to-all-v "
                                "+ unpack, wall time on the slowest rank, clocked from the moment "
                                "the transport launch has finished (a HIP event on its stream); "
                                "transport_wait = host time until that event; collectives = the part "
                                "of exchange spent inside the count all-gather and the all-to-all-v "
                                "(waiting for the slowest peer included)"},
            "kernel_diagnostics": main_stats,
            "roofline": roof,
            "arithmetic": {"mode": md.pkg.arithmetic(),
                           "note": "gray IMC tracking step: 'lean' (default) = face distance by a "
                                   "once-refined reciprocal, fused position update, uncompensated "
                                   "log, each within 4e-15 (relative) of 'exact', whose results equal the CPU "
                                   "oracle's bit for bit; stated tolerance of lean: every particle "
                                   "attribute within 1e-9 PER CYCLE (after one full cycle; 1e-8 after two), "
                                   "integer attributes equal; measured growth about one decade per cycle: 1e-4 of "
                                   "the domain and <= 1e-3 of the histories re-sequenced after ten cycles -- the "
                                   "same as the CPU path's libm vs portable flavours (tests/test_gpu_lean.py; "
                                   "this run's figures: accuracy.lean_vs_exact_after_10_cycles); at any length the "
                                   "tally within 6 sigma of the CPU path's per cell; DDMC steps: exact only",
                           "other_variant": other},
        }
        if md.phase_times is not None:
            out["phase_ms_per_step"] = {k: 1e3 * v / args.steps for k, v in md.phase_times.items()}
        if args.gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.block_nx)
            if not args.no_accuracy:
                out["accuracy"] = accuracy(device, out["cpu_baseline"]["cores"])
                if not md.pkg.Param("use_ddmc"):
                    out["accuracy"]["lean_vs_exact_after_10_cycles"] = lean_vs_exact_after_10_cycles(device)
        print(json.dumps(out), flush=True)
    if comm is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
