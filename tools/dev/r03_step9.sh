set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "ddmc or hybrid" > gpurun_out/s9_pytest.txt 2>&1 || { tail -30 gpurun_out/s9_pytest.txt; exit 1; }
tail -2 gpurun_out/s9_pytest.txt
bash tools/dev/ab2.sh c3 100000000 base cur w4 cur w4 | tee gpurun_out/s9_ab.txt
bash tools/dev/ab2.sh c3-1d 100000000 base cur w4 | tee -a gpurun_out/s9_ab.txt
