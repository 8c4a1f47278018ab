# full check of the tree on the GPU box: pytest -m gpu, then the side benches
set -e
mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > gpurun_out/val_pytest.txt 2>&1 || { tail -40 gpurun_out/val_pytest.txt; exit 1; }
tail -3 gpurun_out/val_pytest.txt
bash tools/dev/ab2.sh c5 10000000 cur | tee gpurun_out/val_ab.txt
bash tools/dev/ab2.sh c2 10000000 cur | tee -a gpurun_out/val_ab.txt
bash tools/dev/ab2.sh c3 100000000 base cur | tee -a gpurun_out/val_ab.txt
