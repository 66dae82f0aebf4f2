#!/bin/bash
# scratch driver of one gpurun call (edited per call; see tools/gpu_ci.sh for the standing steps)
mkdir -p gpurun_out
# ping-pong probe: a bounded first call (a barrier imbalance would hang the workgroup), then timing, then parity
NMV_HIP_LIB=build/abl/lib_PP1.so timeout -k 10 90 python tools/bench_gemm.py --native --ms 64 --shapes gate_up 2>&1 | grep -v amdgpu > gpurun_out/r3_pp1.log; rc=$?; echo "PP1 rc=$rc"; cat gpurun_out/r3_pp1.log
if [ $rc -ne 0 ]; then echo "STOP"; exit 1; fi
NMV_HIP_LIB=build/abl/lib_PP2.so timeout -k 10 90 python tools/bench_gemm.py --native --ms 64 --shapes gate_up 2>&1 | grep -v amdgpu > gpurun_out/r3_pp2.log; rc=$?; echo "PP2 rc=$rc"; cat gpurun_out/r3_pp2.log
if [ $rc -ne 0 ]; then echo "STOP"; exit 1; fi
timeout -k 10 90 python tools/bench_gemm.py --native --ms 64 --shapes gate_up 2>&1 | grep -v amdgpu; echo "(default build above)"
NMV_HIP_LIB=build/abl/lib_PP1.so timeout -k 10 500 python -m pytest tests/test_gpu_w4_native.py tests/test_gpu_w4a16.py -x -q --timeout 120 > gpurun_out/t_pp.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/t_pp.log
