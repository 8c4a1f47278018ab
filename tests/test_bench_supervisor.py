"""bench.py with N > 1: every rank of the launcher supervises a worker process (bench.supervise).

What the first RCCL run on the driver's 8-GPU node must not be able to do is waste the lease: a
worker that dies, a backend that refuses to come up or a collective that never returns has to end
in a labelled fallback or a diagnostic and a non-zero exit, never in a hang.  The worker's process
group life cycle is exercised here on CPU (JB_BENCH_FAKE_WORKER=1: rendezvous over gloo, one
barrier per step, failures injected through the environment); the real worker runs on the GPU box
(tools/dev/rehearse.sh, profiles/r04_rehearsal_*).
"""
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(nranks, extra_env, timeout=150, steps=3):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, JB_BENCH_FAKE_WORKER="1", MASTER_ADDR="127.0.0.1", **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "JB_BENCH_WORKER"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", str(nranks), "--steps", str(steps), "--warmup", "1"]
    t0 = time.time()
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    return res, lines, time.time() - t0


def test_workers_succeed_and_rank0_relays_one_line():
    res, lines, _ = _launch(2, {})
    assert res.returncode == 0, res.stderr[-2000:]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and "backend" not in d   # first attempt (RCCL) went through
    assert "fake worker over nccl" in d["config"]["parallelism"]


def test_rccl_failure_falls_back_to_fresh_gloo_workers_and_says_so():
    res, lines, _ = _launch(2, {"JB_BENCH_FAKE_NCCL_FAILS": "1"})
    assert res.returncode == 0, res.stderr[-2000:]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["backend"].startswith("gloo (rccl failed:") and "exit code 3" in d["backend"]
    assert "fake worker over gloo" in d["config"]["parallelism"]
    assert "RCCL refused" in res.stderr          # the failed attempt's per-rank logs are shown


def test_rank_killed_mid_run_ends_in_a_diagnostic_and_nonzero_exit_not_a_hang():
    # rank 1 dies at step 2 of the (pinned) gloo run: rank 0's worker is then stuck in a barrier
    res, lines, wall = _launch(2, {"JB_BENCH_BACKEND": "gloo", "JB_BENCH_FAKE_KILL": "1@2"})
    assert res.returncode != 0
    assert not lines
    assert wall < 120.0
    # the culprit, not a victim (rank 0's worker breaks on "connection reset by peer" a moment later), is
    # the attempt's reason: first failure wins; every rank's own reason is listed, and the per-rank logs
    # carry the worker's last words
    assert "failed (rank 1: worker exit code 17)" in res.stderr, res.stderr[-3000:]
    assert "rank 1: worker exit code 17" in res.stderr and "dies at step 2" in res.stderr
    assert "[supervisor] worker exit code 17" in res.stderr


def test_hung_rank_is_cut_off_by_the_budget():
    # rank 0's worker never returns: the attempt's time limit (from JB_BENCH_BUDGET_S) ends it
    res, lines, wall = _launch(2, {"JB_BENCH_BACKEND": "gloo", "JB_BENCH_FAKE_HANG": "0@1",
                                   "JB_BENCH_BUDGET_S": "40"})
    assert res.returncode != 0 and not lines
    assert wall < 100.0
    assert "no result after" in res.stderr
