set -e
mkdir -p gpurun_out
bash tools/dev/ab2.sh c3 100000000 cur xextra | tee gpurun_out/r02_c21_ab.txt
