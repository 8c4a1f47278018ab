set -e
mkdir -p gpurun_out
bash tools/dev/ab2.sh c2 10000000 cur fast cur fast | tee gpurun_out/r02_c28_ab.txt
