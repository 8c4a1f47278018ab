set -e
mkdir -p gpurun_out
bash tools/dev/pmc2.sh c3 100000000 cur base | tee gpurun_out/r02_c33_pmc.txt
