set -e
mkdir -p gpurun_out
bash tools/dev/ab2.sh c3 100000000 cur w3 w5 w6 w8 db64 db256 dc64 dc256 | tee gpurun_out/r02_c9_ab.txt
bash tools/dev/ab2.sh c3-1d 100000000 cur w5 w6 w8 | tee -a gpurun_out/r02_c9_ab.txt
