set -e
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lean.py tests/test_gpu_multirank.py tests/test_gpu_regression.py -x -q > gpurun_out/s15_pytest.txt 2>&1 || { tail -40 gpurun_out/s15_pytest.txt; exit 1; }
tail -2 gpurun_out/s15_pytest.txt
bash tools/dev/ab2.sh c3 100000000 cur
bash tools/dev/ab2.sh c3-1d 100000000 cur
