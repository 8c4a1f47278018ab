set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lean.py tests/test_gpu_multirank.py -x -q > gpurun_out/s11_pytest.txt 2>&1 || { tail -30 gpurun_out/s11_pytest.txt; exit 1; }
tail -2 gpurun_out/s11_pytest.txt
bash tools/dev/ab2.sh c5 10000000 prev cur prev cur | tee gpurun_out/s11_ab.txt
