set -e
mkdir -p gpurun_out
bash tools/dev/ab2.sh c2 10000000 cur | tee gpurun_out/r02_c6_ab.txt
bash tools/dev/ab2.sh c3 100000000 r01 cur | tee -a gpurun_out/r02_c6_ab.txt
bash tools/dev/ab2.sh c5 10000000 r01 cur | tee -a gpurun_out/r02_c6_ab.txt
bash tools/dev/pmc2.sh c3 100000000 cur | tee gpurun_out/r02_c6_pmc.txt
