#!/bin/bash
# scratch driver of one gpurun call (edited per call; see tools/gpu_ci.sh for the standing steps)
mkdir -p gpurun_out
timeout -k 10 300 python tools/bench_gemm.py --native --ms 16,64 --shapes qkv70,o70,gate_up70,down70,qkv,o,down 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_gemm70b.log
timeout -k 10 600 python -m pytest tests/test_gpu_w4_native.py tests/test_gpu_w4a16.py -x -q --timeout 120 > gpurun_out/t_w4c.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/t_w4c.log
