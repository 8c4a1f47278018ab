set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lean.py -x -q > gpurun_out/s6_pytest.txt 2>&1 || { tail -40 gpurun_out/s6_pytest.txt; exit 1; }
tail -2 gpurun_out/s6_pytest.txt
bash tools/dev/ab2.sh c5 10000000 cur cur@JB_HYBRID_IMC_BUDGET=128 cur@JB_HYBRID_IMC_BUDGET=192 cur@JB_HYBRID_IMC_BUDGET=256 cur@JB_HYBRID_IMC_BUDGET=64 | tee gpurun_out/s6_ab_c5.txt
