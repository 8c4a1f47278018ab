set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/c29_pytest.txt 2>&1 || { tail -40 gpurun_out/c29_pytest.txt; exit 1; }
tail -2 gpurun_out/c29_pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --no-cpu-baseline > gpurun_out/c29_bench.json
python - <<'P'
import json
d=json.load(open('gpurun_out/c29_bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms_avg'], d['arithmetic']['mode'], d['arithmetic']['other_variant'])
P
