#!/bin/bash
# scratch driver of one gpurun call (edited per call; see tools/gpu_ci.sh for the standing steps)
mkdir -p gpurun_out
timeout -k 10 500 python tools/sweep_stream.py --native --mode 2 --ms 1,16,64 --shapes qkv,o,down 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_sweep_native_m2.log
timeout -k 10 300 python tools/sweep_stream.py --native --mode 1 --ms 1,16,64 --shapes gate_up 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_sweep_native_m1.log
timeout -k 10 200 python bench.py --quant w8a8 --kv-cache-dtype fp8 --no-cpu-baseline > gpurun_out/bench_w8a8.log 2>&1; tail -c 400 gpurun_out/bench_w8a8.log
timeout -k 10 200 python bench.py --quant bf16 --no-cpu-baseline > gpurun_out/bench_bf16.log 2>&1; tail -c 400 gpurun_out/bench_bf16.log
