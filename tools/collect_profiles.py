#!/usr/bin/env python3
"""Turns the rocprofv3 output of `tools/dev/prof.sh <tag>` (gpurun_out/<tag>prof/) into the summaries
committed under profiles/<tag>_*: per-workload PMC summary (raw counter sums of the tracking kernel,
one --pmc pass per group, plus the derived figures quoted in DESIGN.md), the kernel-stats tables
and the bench lines.  One collector for every round: usage: collect_profiles.py <tag> [src_dir]"""
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = sys.argv[2] if len(sys.argv) > 2 else f"gpurun_out/{tag}prof"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
KERNELS = ("k_transport", "k_imc_cell", "k_ddmc_all", "k_ddmc_q", "k_hybrid")
HYBRID_MS = {}


def last_json_line(path):
    lines = [l for l in open(path) if l.startswith("{")]
    return json.loads(lines[-1]) if lines else None


def pmc_pass(d):
    """Counter sums and duration of the tracking kernel(s) of one pass (bench.py --steps 1): the
    kernel with the longest launch names the summary; a hybrid deck runs three k_hybrid launches
    per step (IMC phase, DDMC phase, remainder), whose counters and durations are summed."""
    dur = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if any(k in r["Kernel_Name"] for k in KERNELS):
                dur.setdefault(r["Kernel_Name"], []).append(
                    (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    if not dur:
        return {}, None, None
    main = max(dur, key=lambda n: sum(dur[n]))
    hybrid = "k_hybrid" in main
    keep = [n for n in dur if "k_hybrid" in n] if hybrid else [main]
    tot = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"] in keep:
                # memory-side counters: the whole step (all launches); SQ / GRBM counters: the main
                # launch only (the ratios derived from them describe one kernel)
                whole_step = r["Counter_Name"].startswith(("FETCH", "WRITE", "TCC"))
                if whole_step or r["Kernel_Name"] == main:
                    tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ms = sum(dur[main]) / len(dur[main])
    if hybrid:
        HYBRID_MS[os.path.basename(d)] = {n.split("(")[0]: sum(dur[n]) for n in keep}
    name = main.split("(")[0] + (" (SQ counters: this launch; memory counters: all three k_hybrid launches of the step)" if hybrid else "")
    return tot, ms, name


for wl, particles in (("c2", 10_000_000), ("c2x", 10_000_000), ("c3", 100_000_000), ("c3-1d", 100_000_000),
                      ("c4", 10_000_000), ("c5", 10_000_000)):
    counters, ms_by_pass, kernel = {}, {}, None
    for p in "ABCDEFGHI":
        d = os.path.join(src, f"pmc_{wl}_{p}")
        if not os.path.isdir(d):
            continue
        tot, ms, name = pmc_pass(d)
        counters.update(tot)
        if ms:
            ms_by_pass[p] = ms
            kernel = name
    if not counters:
        continue
    b = last_json_line(os.path.join(src, f"pmc_{wl}_A.json"))
    k = b["kernel_diagnostics"]
    ev, passes, services = k["n_events"], k["n_wave_passes"], k["n_wave_services"]
    ms = ms_by_pass.get("A")
    g = lambda c: counters.get(c, float("nan"))
    clk = g("GRBM_GUI_ACTIVE") / 8 / (ms_by_pass.get("B", ms) * 1e-3)
    hbm = (g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024
    # Share of the SIMDs' cycles in which a VALU instruction was being issued: SQ_ACTIVE_INST_VALU counts
    # quad-cycles summed over the waves, the chip has 1024 SIMDs, the launch lasted GRBM_GUI_ACTIVE / 8 cycles
    # (pass B; scaled to pass A's duration).  (Rounds 2 - 5 took (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES) x
    # SQ_WAVES / 1024, which assumes every wave resident for the whole launch and read 1.02 on configs[3].)
    simd_cycles_a = 1024.0 * (g("GRBM_GUI_ACTIVE") / 8.0) * (ms / ms_by_pass.get("B", ms)) if ms else float("nan")
    valu_busy = min(1.0, 4.0 * g("SQ_ACTIVE_INST_VALU") / simd_cycles_a) if simd_cycles_a == simd_cycles_a else float("nan")
    alg_hbm = (b.get("roofline") or {}).get("algorithmic_hbm_bytes_per_launch")
    summary = {
        "workload": "c2" if wl == "c2x" else wl, "particles_per_gpu": particles, "kernel": kernel,
        "command": "rocprofv3 --kernel-trace --pmc <one group per pass> --output-format csv -- python3 bench.py "
                   + ("" if wl == "c2" else ("--arithmetic exact " if wl == "c2x" else f"--workload {wl} --particles-per-gpu {particles} "))
                   + "--steps 1 --warmup 0 --no-cpu-baseline" + ("" if wl in ("c3", "c3-1d", "c5") else " --no-other-variant"),
        "launch_ms_by_pass": ms_by_pass, "events_per_launch": ev, "wave_passes": passes,
        "service_phases": services,
        "hbm_bytes_per_launch": hbm,
        "hbm_note": "FETCH_SIZE + WRITE_SIZE (KB, separate passes) x 1024, raw: the gfx950 x2 correction of "
                    "FETCH_SIZE is calibrated for 16 B/lane coalesced streams, not for 8-byte gathers",
        "valu_instructions_per_64lane_event": g("SQ_INSTS_VALU") / (ev / 64),
        "valu_instructions_per_wave_pass": g("SQ_INSTS_VALU") / passes,
        "lanes_per_wave_pass": ev / passes,
        "valu_lane_utilisation": g("SQ_THREAD_CYCLES_VALU") / (64 * g("SQ_ACTIVE_INST_VALU")),
        "wave_time_fraction_valu": g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES"),
        "wave_time_fraction_waiting_to_issue": g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"),
        "wave_time_fraction_waiting_on_memory": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"),
        "waves_per_simd": g("SQ_WAVES") / 1024.0,
        "simd_valu_busy_fraction": valu_busy,
        "cycles_per_valu_instruction_per_wave": 4 * g("SQ_ACTIVE_INST_VALU") / g("SQ_INSTS_VALU"),
        "effective_clock_GHz": clk / 1e9,
        "l2_hit_rate": g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")),
        "l2_atomics": g("TCC_EA0_ATOMIC_sum"),
        "f64_fma_add_mul_per_wave_pass": (g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_ADD_F64")
                                          + g("SQ_INSTS_VALU_MUL_F64")) / passes,
        "hbm_GBps": hbm / (ms * 1e-3) / 1e9 if ms else None,
        # the two honest bounds of an instruction-issue-bound kernel, from the counters alone:
        #  * share of the SIMDs' cycles in which a VALU instruction was being issued (4 cycles per
        #    wave-instruction, 16 for the FP64 transcendentals) -- how close the kernel is to the
        #    issue limit FOR THE INSTRUCTION STREAM IT EXECUTES;
        #  * FP64 flop actually executed (fma = 2, add / mul = 1, x 64 lanes x lane utilisation) over
        #    the vector FP64 peak, 78.6 TF/s
        "valu_issue_frac": valu_busy,
        "valu_issue_frac_note": "4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), clamped to 1",
        "algorithmic_hbm_bytes_per_launch": alg_hbm,
        "wasted_traffic_ratio": (hbm / alg_hbm) if alg_hbm else None,
        "fp64_counter_frac": ((2.0 * g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64"))
                              * 64.0 * g("SQ_THREAD_CYCLES_VALU") / (64 * g("SQ_ACTIVE_INST_VALU"))
                              / (ms_by_pass.get("C", ms) * 1e-3) / 78.6e12) if ms else None,
        # L2 (TCC) request path, where collected (passes H, I): busy share of the 128 channels' cycles, requests
        # per channel and cycle -- the guide's gather microbenchmark (MI355X_MICROARCH.md, "Indexed rows": 16.8
        # TB/s of 128-byte lines from the XCDs' L2) is 0.47 -- and the vector L1's requests to L2 with their
        # average round trip in cycles
        "l2_busy_fraction": g("TCC_BUSY_sum") / g("TCC_CYCLE_sum"),
        "l2_requests_per_channel_cycle": g("TCC_REQ_sum") / g("TCC_CYCLE_sum"),
        "l2_read_requests_from_l1": g("TCP_TCC_READ_REQ_sum"),
        "l1_to_l2_read_latency_cycles": g("TCP_TCC_READ_REQ_LATENCY_sum") / g("TCP_TCC_READ_REQ_sum"),
        "non_fp64_valu_per_wave_pass": (g("SQ_INSTS_VALU") - g("SQ_INSTS_VALU_FMA_F64") - g("SQ_INSTS_VALU_ADD_F64")
                                        - g("SQ_INSTS_VALU_MUL_F64") - g("SQ_INSTS_VALU_TRANS_F64")) / passes,
        "counters": counters,
    }
    if any(k_.startswith(f"pmc_{wl}_") for k_ in HYBRID_MS):
        summary["launch_ms_by_kernel_pass_A"] = HYBRID_MS.get(f"pmc_{wl}_A")
        summary["events_note"] = ("events / wave passes / service phases are those of the whole step (three launches); "
                                  "the per-pass figures divide the main launch's SQ counters by them and are upper bounds "
                                  "by the share of the other two launches (~8 % of the time)")
    with open(os.path.join(out, f"{tag}_pmc_summary_" + ("c2_exact" if wl == "c2x" else wl) + ".json"), "w") as fh:
        json.dump(summary, fh, indent=1)
        fh.write("\n")
    print(wl, {k_: (round(v, 4) if isinstance(v, float) else v) for k_, v in summary.items()
               if k_ not in ("counters", "command", "hbm_note", "launch_ms_by_pass")})

for wl in ("c2", "c3", "c5"):
    for f in glob.glob(os.path.join(src, f"stats_{wl}", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(out, f"{tag}_bench_{wl}_kernel_stats.csv"))
    p = os.path.join(src, f"bench_{wl}_under_rocprof.json")
    if os.path.exists(p) and last_json_line(p):
        json.dump(last_json_line(p), open(os.path.join(out, f"{tag}_bench_{wl}_under_rocprof.json"), "w"), indent=1)
for wl in ("c1", "c2", "c2_20steps", "c2_12500k", "c3", "c3-1d", "c4", "c5"):
    p = os.path.join(src, f"bench_{wl}.json")
    if os.path.exists(p) and last_json_line(p):
        json.dump(last_json_line(p), open(os.path.join(out, f"{tag}_bench_{wl}.json"), "w"), indent=1)
print(sorted(f for f in os.listdir(out) if f.startswith(tag)))
