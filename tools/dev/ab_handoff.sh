# usage: ab_handoff.sh <particles per rank> <tag> ...   (tag "cur" = the in-tree library, else variants/libjb_<tag>.so)
# BASELINE configs[4] split by blocks over 2 gloo ranks that share the card: photons handed over per step and the
# time of the hand-off (count kernel, read-back, all-gather, pack kernel, all-to-all-v staged through the host, unpack)
mkdir -p gpurun_out
P=$1; shift
for tag in "$@"; do
  if [ "$tag" = cur ]; then L=$PWD/jaybenne_amd/libjaybenne_amd.so; else L=$PWD/variants/libjb_$tag.so; fi
  JAYBENNE_AMD_LIB=$L JB_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --workload c5 --decomposition blocks --steps 3 --warmup 1 \
      --particles-per-gpu $P --no-cpu-baseline > gpurun_out/abh_$tag.json 2> gpurun_out/abh_$tag.err || { tail -5 gpurun_out/abh_$tag.err; continue; }
  python - gpurun_out/abh_$tag.json $tag <<'P'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
h = d["handoff"]
print(f"{sys.argv[2]:10s} ms/step {d['ms_per_step']:8.2f}  records/step {h['records_per_step']:.0f}  exchange ms {h['exchange_ms_per_step_max_rank']:.2f}  "
      f"collectives ms {h['collectives_ms_per_step_max_rank']:.2f}  iterations/step {d.get('transport_iterations_per_step')}")
P
done
