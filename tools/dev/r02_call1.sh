set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_c1_pytest.txt 2>&1 || { tail -40 gpurun_out/r02_c1_pytest.txt; exit 1; }
tail -3 gpurun_out/r02_c1_pytest.txt
bash tools/dev/ab2.sh c2 10000000 cur cur@JB_NO_EXACT_GEOM=1 noasm b48 b160 b256 | tee gpurun_out/r02_c1_ab.txt
bash tools/dev/ab2.sh c4 10000000 cur | tee -a gpurun_out/r02_c1_ab.txt
bash tools/dev/ab2.sh c5 10000000 cur | tee -a gpurun_out/r02_c1_ab.txt
bash tools/dev/ab2.sh c3 100000000 cur | tee -a gpurun_out/r02_c1_ab.txt
