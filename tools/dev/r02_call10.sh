set -e
mkdir -p gpurun_out
bash tools/dev/pmc2.sh c3 100000000 cur w3 | tee gpurun_out/r02_c10_pmc.txt
JAYBENNE_AMD_LIB=$PWD/variants/libjb_timing.so timeout -k 10 300 python bench.py --workload c3 --particles-per-gpu 100000000 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/timing_c3.json 2> gpurun_out/timing_err.txt
python - gpurun_out/timing_c3.json <<'P'
import json, sys
d = json.load(open(sys.argv[1])); k = d["kernel_diagnostics"]
ev, sv = k["n_wave_passes"] * 1024, k["n_wave_services"] * 1024
print(sys.argv[1], "kernel ms", round(d["roofline"]["kernel_ms_avg"], 2), "wave-cycles in event loop", f"{ev:.3e}", "in service", f"{sv:.3e}",
      "service share", round(sv / (ev + sv), 3), "events", k["n_events"])
P
