# usage: timing.sh <workload> <particles> ...pairs   (variants/libjb_timing.so = a -DJB_TIMING build)
# wave-cycles / 1024 in the event loop / the service phase of k_ddmc_all, and per sub-phase
mkdir -p gpurun_out
while [ $# -ge 2 ]; do
  w=$1; n=$2; shift 2
  env JAYBENNE_AMD_LIB=$PWD/variants/libjb_timing.so timeout -k 10 300 python bench.py --workload $w --particles-per-gpu $n \
      --steps 1 --warmup 1 --no-cpu-baseline --no-other-variant > gpurun_out/timing_$w.json 2> gpurun_out/timing_$w.err
  echo "== $w"; grep JB_TIMING gpurun_out/timing_$w.err | tail -2
  python - gpurun_out/timing_$w.json <<'P'
import json, sys
d = json.load(open(sys.argv[1])); k = d["kernel_diagnostics"]
print("event-loop kcycles", k["n_wave_passes"], "service kcycles", k["n_wave_services"], "ms", d["ms_per_step"])
P
done
