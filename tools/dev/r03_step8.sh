set -e
mkdir -p gpurun_out
bash tools/dev/ab2.sh c3 100000000 base cur base cur | tee gpurun_out/s8_ab_c3.txt
bash tools/dev/ab2.sh c2 10000000 cur | tee -a gpurun_out/s8_ab_c3.txt
