set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_lean.py -x -q > gpurun_out/s4_pytest.txt 2>&1 || { tail -30 gpurun_out/s4_pytest.txt; exit 1; }
tail -2 gpurun_out/s4_pytest.txt
bash tools/dev/ab2.sh c2 10000000 dir2 cur dir2 cur | tee gpurun_out/s4_ab_c2.txt
bash tools/dev/ab2.sh c4 10000000 dir2 cur | tee gpurun_out/s4_ab_c4.txt
bash tools/dev/pmc2.sh c2 10000000 cur > gpurun_out/s4_pmc.txt 2>&1 || true
grep -v "^pmc pass" gpurun_out/s4_pmc.txt | tail -4
