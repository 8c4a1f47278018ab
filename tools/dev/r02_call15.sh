set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_c15_pytest.txt 2>&1 || { tail -40 gpurun_out/r02_c15_pytest.txt; exit 1; }
tail -3 gpurun_out/r02_c15_pytest.txt
bash tools/dev/ab2.sh c3 100000000 cur | tee gpurun_out/r02_c15_ab.txt
bash tools/dev/ab2.sh c3-1d 100000000 cur | tee -a gpurun_out/r02_c15_ab.txt
JAYBENNE_AMD_LIB=$PWD/variants/libjb_timing.so timeout -k 10 300 python bench.py --workload c3 --particles-per-gpu 100000000 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/timing_c3.json 2> gpurun_out/timing_err.txt
grep JB_TIMING gpurun_out/timing_err.txt | tail -1
python - gpurun_out/timing_c3.json <<'P'
import json, sys
d = json.load(open(sys.argv[1])); k = d["kernel_diagnostics"]
ev, sv = k["n_wave_passes"] * 1024, k["n_wave_services"] * 1024
print("kernel ms", round(d["roofline"]["kernel_ms_avg"], 2), "wave-cycles in event loop", f"{ev:.3e}", "in service", f"{sv:.3e}", "service share", round(sv / (ev + sv), 3))
P
