set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/c24_pytest.txt 2>&1 || { tail -30 gpurun_out/c24_pytest.txt; exit 1; }
tail -2 gpurun_out/c24_pytest.txt
bash tools/dev/ab2.sh c5 10000000 base cur | tee gpurun_out/r02_c24_ab.txt
bash tools/dev/ab2.sh c2 10000000 base cur | tee -a gpurun_out/r02_c24_ab.txt
bash tools/dev/ab2.sh c3 100000000 base cur | tee -a gpurun_out/r02_c24_ab.txt
python bench.py --workload c5 --particles-per-gpu 10000000 --no-cpu-baseline --no-accuracy > gpurun_out/r02_bench_c5.json
