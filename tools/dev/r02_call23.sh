set -e
mkdir -p gpurun_out
bash tools/dev/pmc2.sh c5 10000000 cur | tee gpurun_out/r02_c23_pmc.txt
bash tools/dev/pmc2.sh c4 10000000 cur | tee -a gpurun_out/r02_c23_pmc.txt
