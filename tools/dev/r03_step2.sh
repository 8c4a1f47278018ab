set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_lean.py tests/test_gpu_accuracy.py -x -q -s > gpurun_out/s2_pytest.txt 2>&1 || { tail -30 gpurun_out/s2_pytest.txt; exit 1; }
grep "lean sqrt" gpurun_out/s2_pytest.txt; tail -3 gpurun_out/s2_pytest.txt
bash tools/dev/ab2.sh c2 10000000 dir1 cur dir1 cur | tee gpurun_out/s2_ab_c2.txt
bash tools/dev/pmc2.sh c2 10000000 cur > gpurun_out/s2_pmc.txt 2>&1 || true
tail -40 gpurun_out/s2_pmc.txt
