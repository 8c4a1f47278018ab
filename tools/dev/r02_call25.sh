set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py -x -q 2>&1 | tail -3
bash tools/dev/ab2.sh c3 100000000 base cur | tee gpurun_out/r02_c25_ab.txt
bash tools/dev/ab2.sh c3-1d 100000000 base cur | tee -a gpurun_out/r02_c25_ab.txt
