set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "defrag" > gpurun_out/s21_pytest.txt 2>&1 || { tail -40 gpurun_out/s21_pytest.txt; exit 1; }
tail -2 gpurun_out/s21_pytest.txt
for d in 4; do timeout -k 10 400 python tools/dev/steps.py c3 100000000 16 $d 2>/dev/null | tail -1; done
for d in 4; do timeout -k 10 400 python tools/dev/steps.py c2 10000000 12 $d 2>/dev/null | tail -1; done
