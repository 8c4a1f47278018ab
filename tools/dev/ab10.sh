# usage: ab10.sh <workload> <particles> <tag> ...   like ab2.sh with bench.py's default 10 steps behind 2
# warm-up cycles (the sort schedule in its steady state); tag "cur" or variants/libjb_<tag>.so
mkdir -p gpurun_out
w=$1; n=$2; shift 2
for tag in "$@"; do
  if [ "$tag" = cur ]; then L=$PWD/jaybenne_amd/libjaybenne_amd.so; else L=$PWD/variants/libjb_$tag.so; fi
  env JAYBENNE_AMD_LIB=$L timeout -k 10 300 python bench.py --workload $w --particles-per-gpu $n --no-cpu-baseline --no-other-variant > gpurun_out/ab10_${w}_$tag.json 2> gpurun_out/ab10_err.txt
  python - gpurun_out/ab10_${w}_$tag.json $tag <<'P'
import json, sys
d = json.load(open(sys.argv[1])); k = d["kernel_diagnostics"]
print(f"{sys.argv[2]:10s} {d['ms_per_step']:8.2f} ms/step  kernel {d['roofline']['kernel_ms_avg']:8.2f} ms  {d['value']:.4e} hist/s  sorts {d['config'].get('defrag_sorts_in_run')}  passes/ev {64*k['n_wave_passes']/max(k['n_events'],1):.3f}  services {k['n_wave_services']}")
P
done
