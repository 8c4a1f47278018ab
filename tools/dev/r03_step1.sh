# round 3, step 1: the reworked lean kernel -- its tests, then A/B against the last commit
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_lean.py tests/test_gpu_accuracy.py -x -q > gpurun_out/s1_pytest.txt 2>&1 || { tail -30 gpurun_out/s1_pytest.txt; exit 1; }
tail -3 gpurun_out/s1_pytest.txt
bash tools/dev/ab2.sh c2 10000000 base cur base cur | tee gpurun_out/s1_ab_c2.txt
bash tools/dev/ab2.sh c4 10000000 base cur | tee gpurun_out/s1_ab_c4.txt
bash tools/dev/ab2.sh c1 100000 base cur | tee gpurun_out/s1_ab_c1.txt
