set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py -x -q 2>&1 | tail -3
bash tools/dev/ab2.sh c5 10000000 base cur cur@JB_NO_EXACT_GEOM=1 | tee gpurun_out/r02_c22_ab.txt
bash tools/dev/ab2.sh c3 100000000 cur@JB_NO_DDMC_ALL=1 | tee -a gpurun_out/r02_c22_ab.txt
