set -e
mkdir -p gpurun_out
bash tools/dev/ab2.sh c3 100000000 cur l32 l256 cur l32 | tee gpurun_out/r02_c27_ab.txt
