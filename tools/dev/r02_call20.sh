set -e
mkdir -p gpurun_out
JB_BENCH_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 2 --steps 2 --warmup 1 --particles-per-gpu 2000000 > gpurun_out/bench_gpus2_gloo.json 2> gpurun_out/bench_gpus2_gloo.err || { tail -20 gpurun_out/bench_gpus2_gloo.err; exit 1; }
python3 - <<'P'
import json
d = json.loads([l for l in open("gpurun_out/bench_gpus2_gloo.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "n_gpus", "ms_per_step", "transport_iterations_per_step", "handoff")})
print(d["config"]["parallelism"], d["roofline"]["kernel"])
P
timeout -k 10 600 python3 bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || { tail -20 gpurun_out/bench_default.err; exit 1; }
python3 - <<'P'
import json
d = json.loads([l for l in open("gpurun_out/bench_default.json") if l.startswith("{")][-1])
print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "roofline", "cpu_baseline", "accuracy")}, indent=1)[:3000])
P
