#!/bin/bash
# gfx-clock cycles of every microbench_isa kernel (W = 1, 2, 3 launches each): GRBM_GUI_ACTIVE per dispatch
set -e
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ubench_pmc
timeout -k 10 240 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/ubench_pmc -o u -- variants/microbench_isa > gpurun_out/ubench_pmc.txt 2>&1
python3 - <<'P'
import csv, glob, collections
rows = collections.OrderedDict()
for f in glob.glob("gpurun_out/ubench_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
        rows.setdefault(r["Kernel_Name"].split("(")[0], []).append((int(r["Dispatch_Id"]), int(r["Workgroup_Size"]), float(r["Counter_Value"])))
ITER = 4096 * 16
print("kernel               gfx cycles per wave-instruction per SIMD (GRBM_GUI_ACTIVE / 8 XCD): W=1, 2, 3")
for k, v in rows.items():
    by_w = {}
    for d, wg, c in sorted(v):
        by_w.setdefault(wg // 256, []).append(c / 8.0 / (ITER * (wg // 256)))
    print(f"{k:22s}", "  ".join(f"W={w} {min(c):6.2f}" for w, c in sorted(by_w.items())))
P
