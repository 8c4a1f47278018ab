set -e
mkdir -p gpurun_out
python tools/dev/lean_dev.py 1000000 1 | tee gpurun_out/lean_dev.txt
python tools/dev/lean_dev.py 1000000 5 | tee -a gpurun_out/lean_dev.txt
