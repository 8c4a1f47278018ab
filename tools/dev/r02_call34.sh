set -e
mkdir -p gpurun_out
JB_BENCH_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --particles-per-gpu 2000000 --steps 2 --warmup 1 > gpurun_out/c34_gpus2.json 2> gpurun_out/c34_gpus2.err || { tail -20 gpurun_out/c34_gpus2.err; exit 1; }
python - <<'P'
import json
d=json.load(open('gpurun_out/c34_gpus2.json'))
print({k:d[k] for k in ('value','n_gpus','ms_per_step','transport_iterations_per_step')}, d['handoff'], d['config']['parallelism'], d['arithmetic']['mode'])
P
JB_BENCH_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 4 --particles-per-gpu 1000000 --steps 2 --warmup 1 > gpurun_out/c34_gpus4.json 2> gpurun_out/c34_gpus4.err || { tail -20 gpurun_out/c34_gpus4.err; exit 1; }
python - <<'P'
import json
d=json.load(open('gpurun_out/c34_gpus4.json'))
print({k:d[k] for k in ('value','n_gpus','ms_per_step','transport_iterations_per_step')}, d['handoff'], d['config']['blocks_per_gpu'], d['config']['halo_blocks_per_gpu'])
P
