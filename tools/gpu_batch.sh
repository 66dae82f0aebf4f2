#!/bin/bash
# scratch driver of one gpurun call (edited per call; see tools/gpu_ci.sh for the standing steps)
mkdir -p gpurun_out
NMV_HIP_LIB=build/abl/lib_PR1.so timeout -k 10 90 python tools/bench_gemm.py --native --ms 48,64 --shapes gate_up 2>&1 | grep -v amdgpu > gpurun_out/r3_pair1.log; rc=$?; echo "PAIR rc=$rc"; cat gpurun_out/r3_pair1.log
if [ $rc -ne 0 ]; then echo "STOP"; exit 1; fi
NMV_HIP_LIB=build/abl/lib_PR0.so timeout -k 10 90 python tools/bench_gemm.py --native --ms 48,64 --shapes gate_up 2>&1 | grep -v amdgpu; echo "(refactored default above)"
NMV_HIP_LIB=build/abl/lib_PR1.so timeout -k 10 90 python tools/bench_gemm.py --native --ms 64 --shapes gate_up 2>&1 | grep -v amdgpu; echo "(pair again)"
NMV_HIP_LIB=build/abl/lib_PR1.so timeout -k 10 500 python -m pytest tests/test_gpu_w4_native.py tests/test_gpu_w4a16.py -x -q --timeout 120 > gpurun_out/t_pair.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/t_pair.log
