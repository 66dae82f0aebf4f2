#!/bin/bash
# scratch driver of one gpurun call (edited per call; see tools/gpu_ci.sh for the standing steps)
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_w4_native.py -x -q > gpurun_out/t_native.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/t_native.log
echo "== direct (default)"; timeout -k 10 120 python tools/bench_gemm.py --native --ms 1,8,16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_direct.log
echo "== staged (NMV_W4S_DIRECT=0)"; NMV_W4S_DIRECT=0 timeout -k 10 120 python tools/bench_gemm.py --native --ms 1,16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_staged.log
timeout -k 10 400 python tools/sweep_stream.py --native --mode 2 --ms 1,16 --shapes qkv,o,down 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_sweep_direct_m2.log
timeout -k 10 200 python tools/sweep_stream.py --native --mode 1 --ms 1,16 --shapes gate_up 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_sweep_direct_m1.log
