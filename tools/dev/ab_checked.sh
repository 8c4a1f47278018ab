#!/bin/bash
# quick check of the lean kernel (tolerance tests) + A/B of library builds on one workload
# usage: r04_ab.sh <workload> <particles> <item> ...   (items as in ab2.sh)
set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_lean.py -x -q -k "stated_tolerance or million" > gpurun_out/r04_ab_tests.log 2>&1 || { tail -40 gpurun_out/r04_ab_tests.log; exit 1; }
tail -1 gpurun_out/r04_ab_tests.log
bash tools/dev/ab2.sh "$@" 2>&1 | tee gpurun_out/r04_ab.log
