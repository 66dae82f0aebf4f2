#!/bin/bash
# scratch driver of one gpurun call (edited per call; see tools/gpu_ci.sh for the standing steps)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
prof() {  # name, rocprof args..., -- command
  name=$1; shift
  timeout -k 10 300 rocprofv3 "$@" > gpurun_out/$name.log 2>&1
  echo "== $name rc=$?"
}
prof gprof --kernel-trace --output-format csv -d gpurun_out/gprof -- python tools/bench_gemm.py --ms 1,16,64
find gpurun_out/gprof -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} gpurun_out/gemm_trace.csv; rm -rf gpurun_out/gprof
prof gprofn --kernel-trace --output-format csv -d gpurun_out/gprofn -- python tools/bench_gemm.py --native --ms 1,16,64
find gpurun_out/gprofn -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} gpurun_out/gemm_trace_native.csv; rm -rf gpurun_out/gprofn
prof gpmc --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/gpmc -- python tools/bench_gemm.py --native --ms 1,64 --shapes gate_up
find gpurun_out/gpmc -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} gpurun_out/gemm_pmc.csv; rm -rf gpurun_out/gpmc
for c in FETCH_SIZE WRITE_SIZE; do
  prof traffic_$c --pmc $c --kernel-trace --output-format csv -d gpurun_out/tr_$c -- python tools/bench_gemm.py --ms 1,16,32,64
  find gpurun_out/tr_$c -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} gpurun_out/traffic_$c.csv; rm -rf gpurun_out/tr_$c
done
ls -la gpurun_out/*.csv
