set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/s5_pytest.txt 2>&1 || { tail -40 gpurun_out/s5_pytest.txt; exit 1; }
tail -3 gpurun_out/s5_pytest.txt
bash tools/dev/ab2.sh c5 10000000 base cur | tee gpurun_out/s5_ab_c5.txt
