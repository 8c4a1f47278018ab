#!/usr/bin/env python3
"""Sums the rocprofv3 --pmc counters of the k_transport launch(es) in gpurun_out/pmc2_<w>_<name>_{A,B,C}
and prints the derived per-pass / per-event figures (see profiles/README.md for the formulas)."""
import csv, glob, json, os, sys
w, name = sys.argv[1], sys.argv[2]
tot, dur, per_pass = {}, [], {}
for p in "ABC":
    d = f"gpurun_out/pmc2_{w}_{name}_{p}"
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if not any(k_ in r["Kernel_Name"] for k_ in ("k_transport", "k_ddmc_all", "k_ddmc_q", "k_imc_cell", "k_hybrid")):
                continue
            per_pass.setdefault(r["Counter_Name"], {}).setdefault(p, 0.0)
            per_pass[r["Counter_Name"]][p] += float(r["Counter_Value"])
            tot["_vgpr"] = r.get("VGPR_Count") or r.get("Arch_VGPR_Count")
            tot["_kernel"] = r["Kernel_Name"][:60]
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if any(k_ in r["Kernel_Name"] for k_ in ("k_transport", "k_ddmc_all", "k_ddmc_q", "k_imc_cell", "k_hybrid")):
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
# a counter collected in several passes (SQ_INSTS_VALU): the mean over those passes
for c, by_pass in per_pass.items():
    tot[c] = sum(by_pass.values()) / len(by_pass)
b = json.load(open(f"gpurun_out/pmc2_{w}_{name}_A.json"))
k = b["kernel_diagnostics"]
ev, passes = k["n_events"], k["n_wave_passes"]
dur = [d for d in dur if d > 0.05 * max(dur)] if dur else dur
ms = sum(dur) / max(len(dur), 1)
out = {"workload": w, "lib": name, "kernel": tot.get("_kernel"), "vgpr": tot.get("_vgpr"), "launch_ms": ms,
       "events": ev, "wave_passes": passes, "services": k["n_wave_services"],
       "counters": {c: v for c, v in tot.items() if not c.startswith("_")}}
g = lambda c: tot.get(c, float("nan"))
clk = g("GRBM_GUI_ACTIVE") / 8 / (ms * 1e-3)                      # Hz
simd_cycles = 1024 * clk * ms * 1e-3
out["derived"] = {
    "valu_per_64lane_event": g("SQ_INSTS_VALU") / (ev / 64),
    "valu_per_wave_pass": g("SQ_INSTS_VALU") / passes,
    "lanes_per_pass": ev / passes,
    "lane_utilisation_valu": g("SQ_THREAD_CYCLES_VALU") / (64 * g("SQ_ACTIVE_INST_VALU")),
    "effective_clock_GHz": clk / 1e9,
    "valu_pipe_busy_frac (SQ_INST_CYCLES_VALU / SIMD-cycles)": g("SQ_INST_CYCLES_VALU") / simd_cycles,
    "wave_valu_active_frac (4 SQ_ACTIVE_INST_VALU / 4 SQ_WAVE_CYCLES)": g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES"),
    "wave_wait_issue_frac": g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"),
    "wave_wait_mem_frac": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"),
    "cycles_per_valu_inst (per wave)": 4 * g("SQ_ACTIVE_INST_VALU") / g("SQ_INSTS_VALU"),
    "salu_per_wave_pass": g("SQ_INSTS_SALU") / passes,
    "simd_cycles_per_wave_pass": simd_cycles / passes,
    "f64_ops_per_pass": (g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64")) / passes,
    "trans_f64_per_pass": g("SQ_INSTS_VALU_TRANS_F64") / passes,
    "int32_per_pass": g("SQ_INSTS_VALU_INT32") / passes, "int64_per_pass": g("SQ_INSTS_VALU_INT64") / passes,
    "cvt_per_pass": g("SQ_INSTS_VALU_CVT") / passes,
}
json.dump(out, open(f"gpurun_out/pmc2_{w}_{name}.json", "w"), indent=1)
print(name, json.dumps({k_: (round(v, 4) if isinstance(v, float) else v) for k_, v in out["derived"].items()}), "ms", round(ms, 2), "vgpr", out["vgpr"])
