# full check of the tree on the GPU box: pytest -m gpu, smoke(), the default bench line
set -e
mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > gpurun_out/val_pytest.txt 2>&1 || { tail -40 gpurun_out/val_pytest.txt; exit 1; }
tail -3 gpurun_out/val_pytest.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 600 python bench.py > gpurun_out/val_bench.json 2> gpurun_out/val_bench.err
python - <<'P'
import json
d = json.loads([l for l in open("gpurun_out/val_bench.json") if l.startswith("{")][-1])
print("bench: value", d["value"], "ms/step", d["ms_per_step"], "kernel", d["roofline"]["kernel_ms_avg"], "frac", d["roofline"]["frac"],
      "traffic", d["roofline"]["traffic"], "other", d["arithmetic"]["other_variant"]["value"], "cpu", d["cpu_baseline"]["value"], "acc", d["accuracy"]["gpu_error"])
P
