# usage: ab1.sh <workload> <particles> <tag> ...   like ab2.sh, but ONE step and no warm-up (for
# timing-only experiment builds whose results are wrong, so that every item sees the same first cycle)
mkdir -p gpurun_out
w=$1; n=$2; shift 2
for tag in "$@"; do
  if [ "$tag" = cur ]; then L=$PWD/jaybenne_amd/libjaybenne_amd.so; else L=$PWD/variants/libjb_$tag.so; fi
  for rep in 1 2; do
  env JAYBENNE_AMD_LIB=$L timeout -k 10 300 python bench.py --workload $w --particles-per-gpu $n \
      --steps 1 --warmup 0 --no-cpu-baseline --no-other-variant > gpurun_out/ab1_${w}_$tag.json 2> gpurun_out/ab1_err.txt
  python - gpurun_out/ab1_${w}_$tag.json $tag <<'P'
import json, sys
d = json.load(open(sys.argv[1])); k = d["kernel_diagnostics"]
print(f"{sys.argv[2]:12s} kernel {d['roofline']['kernel_ms_avg']:8.2f} ms  events {k['n_events']}  passes {k['n_wave_passes']}  services {k['n_wave_services']}")
P
  done
done
