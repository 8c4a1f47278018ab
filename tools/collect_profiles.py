#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of tools/dev/prof_r01*.sh (under gpurun_out/) into the small
summaries committed under profiles/: per-pass PMC sums for the transport kernel, the kernel-stats
table, the bench lines, and the HBM-traffic file bench.py reads for roofline.traffic.

usage: collect_profiles.py <tag, e.g. r01_d> <gpurun_out dir>"""
import csv
import glob
import json
import os
import shutil
import sys

tag, src = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
rnd = tag.split("_")[1]                       # "d"


def pmc(dirname, command):
    files = glob.glob(os.path.join(src, dirname, "*counter_collection.csv")) + \
        glob.glob(os.path.join(src, dirname, "*", "*counter_collection.csv"))
    if not files:
        return None
    rows = [r for r in csv.DictReader(open(files[0])) if "k_transport" in r["Kernel_Name"]]
    c = {}
    for r in rows:
        c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    dur = (int(rows[0]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) * 1e-6
    return {"kernel": rows[0]["Kernel_Name"].split("(")[0], "command": command, "launch_ms": dur,
            "vgpr": int(rows[0]["VGPR_Count"]), "counters": c}


def write(name, obj):
    with open(os.path.join(out, f"{tag}_{name}.json"), "w") as fh:
        json.dump(obj, fh, indent=1)
        fh.write("\n")


for wl, sfx, cmd in (("c2", "", "python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline"),
                     ("c3", "_c3", "python3 bench.py --workload c3 --particles-per-gpu 100000000 "
                                   "--steps 1 --warmup 0 --no-cpu-baseline")):
    got = {}
    for d in sorted(glob.glob(os.path.join(src, f"pmc_r1{rnd}{sfx}_*"))):
        name = os.path.basename(d)[len(f"pmc_r1{rnd}{sfx}_"):]
        if wl == "c2" and name.startswith("c3"):
            continue
        p = pmc(os.path.basename(d), f"rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- {cmd}")
        if p:
            p["workload"] = wl
            write(f"pmc_{wl}_{name}", p)
            got[name] = p
    if "FETCH_SIZE" in got and "WRITE_SIZE" in got:
        f = got["FETCH_SIZE"]["counters"]["FETCH_SIZE"]
        w = got["WRITE_SIZE"]["counters"]["WRITE_SIZE"]
        write(f"hbm_traffic_{wl}", {
            "workload": wl, "particles_per_gpu": 10_000_000 if wl == "c2" else 100_000_000,
            "kernel": got["FETCH_SIZE"]["kernel"], "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
            "hbm_bytes_per_launch": (f + w) * 1024,
            "note": "separate --pmc passes; rocprofv3 reports KB.  The gfx950 x2 correction of "
                    "FETCH_SIZE applies to wide (16 B/lane) coalesced streams; this kernel reads "
                    "8-byte gathers and 8-byte particle fields, which the guide lists as "
                    "uncalibrated, so the raw value is used (a lower bound within 2x)."})
    for stats in glob.glob(os.path.join(src, f"prof_r1{rnd}{sfx}", "*kernel_stats.csv")):
        shutil.copy(stats, os.path.join(out, f"{tag}_bench_{wl}_kernel_stats.csv"))
for a, b in ((f"bench_prof_{rnd}.json", "bench_c2_under_rocprof"), (f"bench_prof_{rnd}_c3.json", "bench_c3_under_rocprof"),
             (f"bench_final_{rnd}.json", "bench")):
    pth = os.path.join(src, a)
    if os.path.exists(pth):
        line = [l for l in open(pth) if l.startswith("{")][-1]
        write(b, json.loads(line))
print(sorted(os.listdir(out)))
