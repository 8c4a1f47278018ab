set -e
python -m pytest tests/test_gpu_lean.py tests/test_gpu_parity.py -x -q 2>&1 | tail -5
bash tools/dev/ab2.sh c5 10000000 cur cur@JB_EXACT_ARITH=1 | tee gpurun_out/r02_c35_ab.txt
