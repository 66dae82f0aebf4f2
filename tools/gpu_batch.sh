#!/bin/bash
# scratch driver of one gpurun call (edited per call; see tools/gpu_ci.sh for the standing steps)
mkdir -p gpurun_out
timeout -k 10 200 python tools/bench_gemm.py --ms 128,256,512,2048 2>&1 | grep -v amdgpu.ids > gpurun_out/r3_gemm_prefill_a2.log; cat gpurun_out/r3_gemm_prefill_a2.log
timeout -k 10 120 python tools/profile_prefill.py > gpurun_out/r3_prefill_a2.log 2>&1; cat gpurun_out/r3_prefill_a2.log
timeout -k 10 800 python -m pytest tests/test_gpu_w4a16.py tests/test_gpu_model.py tests/test_gpu_wq_formats.py tests/test_gpu_linear_methods.py -q -m gpu --timeout 600 > gpurun_out/r3_t_sub.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r3_t_sub.log
