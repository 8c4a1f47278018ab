# full validation of the committed tree on one MI355X box
set -e
mkdir -p gpurun_out
make -C examples mpi > gpurun_out/val_mpi_build.txt 2>&1 || true
python -m pytest tests -m gpu -x -q > gpurun_out/val_pytest.txt 2>&1
tail -3 gpurun_out/val_pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/val_bench.json 2> gpurun_out/val_bench.err
cat gpurun_out/val_bench.json
