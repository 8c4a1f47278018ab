#!/bin/bash
# first GPU check of k_imc_cell: lean-tolerance + accuracy + parity tests, then A/B on C2 against round 3's library
set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_lean.py tests/test_gpu_accuracy.py -x -q > gpurun_out/r04_first_tests.log 2>&1 || { tail -40 gpurun_out/r04_first_tests.log; exit 1; }
tail -3 gpurun_out/r04_first_tests.log
bash tools/dev/ab2.sh c2 10000000 r03 cur cur@JB_NO_IMC_CELL=1 2>&1 | tee gpurun_out/r04_first_ab.log
