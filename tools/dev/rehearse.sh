#!/bin/bash
# Multi-rank rehearsals of bench.py on the one-GPU box (ranks share the card; records travel over gloo):
#  1. --gpus 2 as the driver types it: the RCCL attempt cannot come up on one card (rank 1 has no
#     device) -> the supervisors fall back to fresh gloo workers and label the line
#  2. C2 on 4 gloo ranks (transport iterations per step, hand-off volume)
#  3. a rank killed mid-run: diagnostic + non-zero exit, wall time recorded
#  4. N = 1 at the north-star's own size (1.25e7 photons per GPU)
mkdir -p gpurun_out
P=${1:-1000000}
t0=$(date +%s)
python bench.py --gpus 2 --steps 2 --warmup 1 --particles-per-gpu $P > gpurun_out/${TAG:-r05}_rehearsal_c2_fallback2.json 2> gpurun_out/${TAG:-r05}_rehearsal_c2_fallback2.err
echo "1. fallback run: rc $? in $(( $(date +%s) - t0 )) s"; tail -c 600 gpurun_out/${TAG:-r05}_rehearsal_c2_fallback2.json | head -c 400; echo
t0=$(date +%s)
JB_BENCH_BACKEND=gloo python bench.py --gpus 4 --steps 2 --warmup 1 --particles-per-gpu $P > gpurun_out/${TAG:-r05}_rehearsal_c2_gloo4.json 2> gpurun_out/${TAG:-r05}_rehearsal_c2_gloo4.err
echo "2. 4 gloo ranks: rc $? in $(( $(date +%s) - t0 )) s"
t0=$(date +%s)
JB_BENCH_BACKEND=gloo JB_BENCH_KILL_RANK=1@0 python bench.py --gpus 2 --steps 2 --warmup 1 --particles-per-gpu $P > gpurun_out/${TAG:-r05}_rehearsal_c2_kill.json 2> gpurun_out/${TAG:-r05}_rehearsal_c2_kill.err
echo "3. killed rank: rc $? in $(( $(date +%s) - t0 )) s" | tee gpurun_out/${TAG:-r05}_rehearsal_c2_kill.txt
grep -c . gpurun_out/${TAG:-r05}_rehearsal_c2_kill.err; tail -12 gpurun_out/${TAG:-r05}_rehearsal_c2_kill.err
# 5. the SMR decks on 4 gloo ranks: BASELINE configs[4] (3 levels, hybrid) with the decomposition bench.py
#    picks (replicated mesh) and with the block partition; configs[3] (block partition, hand-off)
for spec in "c5 auto ${TAG:-r05}_rehearsal_c5_gloo4" "c5 blocks ${TAG:-r05}_rehearsal_c5_gloo4_blocks" "c4 auto ${TAG:-r05}_rehearsal_c4_gloo4"; do
  set -- $spec
  t0=$(date +%s)
  JB_BENCH_BACKEND=gloo python bench.py --gpus 4 --workload $1 --decomposition $2 --steps 2 --warmup 1 --particles-per-gpu $P > gpurun_out/$3.json 2> gpurun_out/$3.err
  echo "5. $1 on 4 gloo ranks ($2): rc $? in $(( $(date +%s) - t0 )) s"
done
python bench.py --particles-per-gpu 12500000 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG:-r05}_bench_c2_12500k.json 2> gpurun_out/${TAG:-r05}_bench_c2_12500k.err
echo "4. N=1 at 1.25e7: rc $?"
python - <<'P'
import json
import os
tag = os.environ.get("TAG", "r05")
for f in (f"{tag}_rehearsal_c2_fallback2", f"{tag}_rehearsal_c2_gloo4", f"{tag}_bench_c2_12500k", f"{tag}_rehearsal_c5_gloo4", f"{tag}_rehearsal_c5_gloo4_blocks", f"{tag}_rehearsal_c4_gloo4"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
        c = d["config"]
        print(f, d["n_gpus"], d.get("backend"), "%.3e" % d["value"], d["ms_per_step"], d.get("transport_iterations_per_step"), d["handoff"]["records_per_step"],
              c.get("decomposition"), c.get("photons_per_rank_min"), c.get("photons_per_rank_max"), c.get("estimated_work_per_rank_max_over_mean"))
    except Exception as e:
        print(f, "no line:", e)
P
