set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_regression.py -m gpu -x -q -k "mpi_handoff or hdf5_dump" > gpurun_out/r02_c19_pytest.txt 2>&1 || { tail -60 gpurun_out/r02_c19_pytest.txt; exit 1; }
tail -3 gpurun_out/r02_c19_pytest.txt
/opt/conda/bin/mpiexec -n 3 examples/handoff_mpi 16 8 200000 3 | tail -5
