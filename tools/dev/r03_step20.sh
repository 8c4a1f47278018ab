set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "ddmc or hybrid or general or faces or lean_kernel" > gpurun_out/s20_pytest.txt 2>&1 || { tail -40 gpurun_out/s20_pytest.txt; exit 1; }
tail -2 gpurun_out/s20_pytest.txt
bash tools/dev/ab2.sh c3-1d 100000000 prev cur prev cur
