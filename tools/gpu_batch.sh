#!/bin/bash
# scratch driver of one gpurun call (edited per call; see tools/gpu_ci.sh for the standing steps)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_tp.py tests/test_gpu_model.py tests/test_gpu_model_golden.py tests/test_gpu_linear_methods.py tests/test_gpu_checkpoints.py -q -m gpu --timeout 400 > gpurun_out/t_tp_model.log 2>&1; echo "rc=$?"; tail -6 gpurun_out/t_tp_model.log
