#!/bin/bash
set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_lean.py tests/test_gpu_regression.py tests/test_gpu_accuracy.py -x -q > gpurun_out/r04_lvl_tests.log 2>&1 || { tail -40 gpurun_out/r04_lvl_tests.log; exit 1; }
tail -1 gpurun_out/r04_lvl_tests.log
bash tools/dev/ab2.sh c4 10000000 cur
bash tools/dev/ab2.sh c5 10000000 cur
bash tools/dev/ab2.sh c2 10000000 cur
