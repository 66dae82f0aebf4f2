#!/bin/bash
# scratch driver of one gpurun call (edited per call; see tools/gpu_ci.sh for the standing steps)
mkdir -p gpurun_out
timeout -k 10 300 python tools/bench_gemm.py --native --ms 1,16,64 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_gemm_final.log
