set -e
python -m pytest tests/test_gpu_parity.py -x -q -k "more_resident" 2>&1 | tail -15
