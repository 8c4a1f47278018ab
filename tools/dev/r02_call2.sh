set -e
bash tools/dev/ab2.sh c2 10000000 r01 noasm cur@JB_NO_EXACT_GEOM=1 r01 | tee gpurun_out/r02_c2_ab.txt
bash tools/dev/pmc2.sh c2 10000000 r01 noasm | tee gpurun_out/r02_c2_pmc.txt
