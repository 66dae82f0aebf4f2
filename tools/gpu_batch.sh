#!/bin/bash
# scratch driver of one gpurun call (edited per call; see tools/gpu_ci.sh for the standing steps)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/tr_$c -- python tools/bench_gemm.py --ms 1,16,32,64 > gpurun_out/traffic_$c.log 2>&1; echo "== $c rc=$?"
  find gpurun_out/tr_$c -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} gpurun_out/traffic_$c.csv; rm -rf gpurun_out/tr_$c
done
timeout -k 10 300 python bench.py > gpurun_out/bench_final.log 2>&1; echo "bench rc=$?"; tail -c 700 gpurun_out/bench_final.log
