set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_c16_pytest.txt 2>&1 || { tail -40 gpurun_out/r02_c16_pytest.txt; exit 1; }
tail -3 gpurun_out/r02_c16_pytest.txt
bash tools/dev/ab2.sh c3 100000000 cur w3 w5 | tee gpurun_out/r02_c16_ab.txt
bash tools/dev/ab2.sh c3-1d 100000000 cur w5 | tee -a gpurun_out/r02_c16_ab.txt
