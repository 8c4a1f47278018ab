# multi-rank rehearsals on ONE GPU over gloo (ranks share the card; production is RCCL, one GPU per rank)
set -e
mkdir -p gpurun_out
export JB_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 400 python bench.py --gpus 2 --steps 2 --warmup 1 --particles-per-gpu 1000000 --no-cpu-baseline > gpurun_out/rehearse_c2_gpus2.json 2> gpurun_out/rehearse_err.txt
timeout -k 10 400 python bench.py --workload c4 --gpus 4 --steps 2 --warmup 1 --particles-per-gpu 1000000 --no-cpu-baseline > gpurun_out/rehearse_c4_gpus4.json 2>> gpurun_out/rehearse_err.txt
timeout -k 10 400 python bench.py --workload c5 --gpus 2 --steps 2 --warmup 1 --particles-per-gpu 1000000 --no-cpu-baseline > gpurun_out/rehearse_c5_gpus2.json 2>> gpurun_out/rehearse_err.txt
python - <<'P'
import json
for f in ("c2_gpus2", "c4_gpus4", "c5_gpus2"):
    d = json.load(open(f"gpurun_out/rehearse_{f}.json"))
    print(f, "ms/step", round(d["ms_per_step"], 2), "iterations/step", d["transport_iterations_per_step"], "handoff", d["handoff"]["records_per_step"],
          "exchange ms", round(d["handoff"]["exchange_ms_per_step_max_rank"], 3), "wait ms", round(d["handoff"]["transport_wait_ms_per_step_max_rank"], 3), "collectives ms", round(d["handoff"]["collectives_ms_per_step_max_rank"], 3))
P
