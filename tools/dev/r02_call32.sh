bash tools/dev/pmc3.sh c3 100000000 | tee gpurun_out/r02_c32_pmc3.txt
