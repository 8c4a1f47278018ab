# lean / parity / accuracy tests, then C2 (and C5, C4) of the in-tree library against variants/libjb_prev.so
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lean.py tests/test_gpu_accuracy.py -x -q > gpurun_out/abc2_pytest.txt 2>&1 || { tail -40 gpurun_out/abc2_pytest.txt; exit 1; }
tail -2 gpurun_out/abc2_pytest.txt
bash tools/dev/ab2.sh c2 10000000 prev cur prev cur
bash tools/dev/ab2.sh c5 10000000 prev cur
