# usage: ab2.sh <workload> <particles> <item> ...   item = tag or tag@ENV=VAL[,ENV=VAL]
#   tag "cur" = the in-tree library, else variants/libjb_<tag>.so; output gpurun_out/ab_<workload>_<item>.json
set -e
mkdir -p gpurun_out
w=$1; n=$2; shift 2
for item in "$@"; do
  tag=${item%%@*}
  envs=""
  if [ "$tag" != "$item" ]; then envs=$(echo "${item#*@}" | tr ',' ' '); fi
  if [ "$tag" = cur ]; then L=$PWD/jaybenne_amd/libjaybenne_amd.so; else L=$PWD/variants/libjb_$tag.so; fi
  out=gpurun_out/ab_${w}_$(echo "$item" | tr '@=,' '___').json
  env JAYBENNE_AMD_LIB=$L $envs timeout -k 10 300 python bench.py --workload $w --particles-per-gpu $n \
      --steps 3 --warmup 1 --no-cpu-baseline > $out 2> gpurun_out/ab_err.txt
  python - "$out" "$item" <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
k = d["kernel_diagnostics"]
print(f"{sys.argv[2]:28s} {d['ms_per_step']:8.2f} ms/step  kernel {d['roofline']['kernel_ms_avg']:8.2f} ms  "
      f"{d['value']:.4e} hist/s  passes/ev {64*k['n_wave_passes']/max(k['n_events'],1):.3f}  "
      f"services {k['n_wave_services']}  {d['roofline']['kernel']}")
P
done
