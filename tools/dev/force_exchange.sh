# the fixed cost of one hand-off iteration in a one-rank group (nothing moves): through the C call over an RCCL
# communicator of the library's own, over torch.distributed callbacks, and driven from Python; RCCL and gloo
set -e
mkdir -p gpurun_out
for spec in "nccl c" "nccl c-torch" "nccl python" "gloo c" "gloo python"; do
  set -- $spec
  JB_BENCH_BACKEND=$1 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-variant --force-exchange --handoff $2 > gpurun_out/fx_$1_$2.json 2> gpurun_out/fx_err.txt || { tail -5 gpurun_out/fx_err.txt; }
done
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-variant > gpurun_out/fx_none.json 2>> gpurun_out/fx_err.txt
python - <<'P'
import json
for f in ("nccl_c", "nccl_c-torch", "nccl_python", "gloo_c", "gloo_python", "none"):
    try: d = json.loads([l for l in open(f"gpurun_out/fx_{f}.json") if l.startswith("{")][-1])   # (RCCL prints its banner on stdout)
    except Exception as e: print(f, "failed", e); continue
    print(f, "ms/step", round(d["ms_per_step"], 3), "kernel", round(d["roofline"]["kernel_ms_avg"], 3), "iter/step", d["transport_iterations_per_step"],
          "exchange ms", round(d["handoff"]["exchange_ms_per_step_max_rank"], 3), "coll ms", round(d["handoff"]["collectives_ms_per_step_max_rank"], 3), "|", d["handoff"].get("path"))
P
