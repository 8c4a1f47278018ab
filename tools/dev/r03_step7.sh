set -e
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lean.py tests/test_gpu_accuracy.py -x -q -s > gpurun_out/s7_pytest.txt 2>&1 || { tail -40 gpurun_out/s7_pytest.txt; exit 1; }
grep -i "photons whose\|largest position\|3-D plane\|lean sqrt" gpurun_out/s7_pytest.txt; tail -3 gpurun_out/s7_pytest.txt
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-variant > gpurun_out/s7_bench.json 2> gpurun_out/s7_err.txt
python - <<'P'
import json
d = json.loads([l for l in open("gpurun_out/s7_bench.json") if l.startswith("{")][-1])
print("c2 ms/step", d["ms_per_step"], "kernel", d["roofline"]["kernel_ms_avg"], "wall-kernel", d["ms_per_step"] - d["roofline"]["kernel_ms_avg"])
P
