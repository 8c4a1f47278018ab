#!/bin/bash
set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_lean.py tests/test_gpu_regression.py -x -q > gpurun_out/r04_hyb_tests.log 2>&1 || { tail -40 gpurun_out/r04_hyb_tests.log; exit 1; }
tail -1 gpurun_out/r04_hyb_tests.log
bash tools/dev/ab2.sh c5 10000000 cur cur@JB_NO_IMC_CELL=1
