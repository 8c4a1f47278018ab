set -e
python -m pytest tests/test_gpu_multirank.py -x -q -k "lean" 2>&1 | tail -5
