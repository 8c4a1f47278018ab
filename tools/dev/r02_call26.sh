set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/c26_pytest.txt 2>&1 || { tail -40 gpurun_out/c26_pytest.txt; exit 1; }
tail -2 gpurun_out/c26_pytest.txt
