set -e
timeout -k 10 600 python -m pytest tests/test_gpu_custom_allreduce.py tests/test_gpu_tp.py -x -q -m gpu --timeout 400 2>&1 | tail -4
