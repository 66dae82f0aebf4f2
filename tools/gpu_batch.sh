#!/bin/bash
# scratch driver of one gpurun call (edited per call; see tools/gpu_ci.sh for the standing steps)
mkdir -p gpurun_out
for rep in 1 2; do for v in Q0 Q1 Q3; do echo "== alternating s_setprio ${v#Q} (0 = none)"; NMV_HIP_LIB=build/abl/lib_$v.so python tools/bench_gemm.py --native --ms 64 --shapes gate_up 2>&1 | grep gate_up; done; done | tee gpurun_out/r3_prio.log
