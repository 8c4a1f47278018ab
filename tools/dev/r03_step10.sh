set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lean.py -x -q > gpurun_out/s10_pytest.txt 2>&1 || { tail -30 gpurun_out/s10_pytest.txt; exit 1; }
tail -2 gpurun_out/s10_pytest.txt
bash tools/dev/ab2.sh c2 10000000 prev cur prev cur | tee gpurun_out/s10_ab.txt
bash tools/dev/ab2.sh c4 10000000 prev cur | tee -a gpurun_out/s10_ab.txt
