#!/bin/bash
# Runs the GPU steps one after another on the gpurun box, each under its own timeout, logging to
# gpurun_out/.  A step that times out or is killed stops the chain (no further GPU step);
# ordinary test failures do not.
mkdir -p gpurun_out
TMO=${TMO:-900}
run() {
  name=$1; shift
  timeout -k 10 "$TMO" "$@" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "== $name rc=$rc"; tail -n "${TAILN:-4}" "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "STOP: $name timed out / was killed"; exit 1; fi
}
for step in "$@"; do
  case "$step" in
    info)   run info bash -c 'rocminfo | grep -E "Marketing|Compute Unit|Max Clock" | head -8; nproc; lscpu | grep "Model name"; free -g | head -2' ;;
    smoke)  run smoke python -c 'import __graft_entry__ as g; g.smoke()' ;;
    cache)  run t_cache python -m pytest tests/test_gpu_cache.py -q -m gpu -x --timeout 120 ;;
    glue)   run t_glue python -m pytest tests/test_gpu_glue.py -q -m gpu --timeout 120 ;;
    attn)   run t_attn python -m pytest tests/test_gpu_attention.py -q -m gpu --timeout 180 ;;
    model)  run t_model python -m pytest tests/test_gpu_model.py -q -m gpu --timeout 300 ;;
    benchtiny) run benchtiny python bench.py --model tiny --steps 16 --warmup 2 --batch 8 --context 64 ;;
    w8)     run t_w8 python -m pytest tests/test_gpu_w8a8.py -q -m gpu --timeout 300 ;;
    wq)     run t_wq python -m pytest tests/test_gpu_wq_formats.py -q -m gpu --timeout 300 ;;
    lm)     run t_lm python -m pytest tests/test_gpu_linear_methods.py -q -m gpu --timeout 300 ;;
    w4)     run t_w4 python -m pytest tests/test_gpu_w4a16.py -q -m gpu --timeout 300 ;;
    all)    run t_all python -m pytest tests -q -m gpu --timeout 300 ;;
    bench)  run bench python bench.py ;;
    gemm)   run gemm python tools/bench_gemm.py ;;
    gemmprof) cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
            run gemmprof rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gprof -- python tools/bench_gemm.py ${GEMM_ARGS:-}
            find gpurun_out/gprof -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} gpurun_out/gemm_trace.csv ;;
    gemmpmc) cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
            run gemmpmc rocprofv3 --pmc ${PMC:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU} --kernel-trace --output-format csv -d gpurun_out/gpmc -- python tools/bench_gemm.py ${GEMM_ARGS:-}
            find gpurun_out/gpmc -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} gpurun_out/gemm_pmc.csv ;;
    traffic) cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
            # HBM bytes of the GEMM launches: FETCH_SIZE and WRITE_SIZE do not fit one pass
            for c in FETCH_SIZE WRITE_SIZE; do
              run traffic_$c rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/tr_$c -- python tools/bench_gemm.py --ms ${TRAFFIC_MS:-1,16,32,64}
              find gpurun_out/tr_$c -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} gpurun_out/traffic_$c.csv
              rm -rf gpurun_out/tr_$c
            done ;;
    attnprof) cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
            run attnprof rocprofv3 --kernel-trace --output-format csv -d gpurun_out/aprof -- python tools/bench_attn.py ${ATTN_ARGS:-}
            find gpurun_out/aprof -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} gpurun_out/attn_trace.csv
            rm -rf gpurun_out/aprof ;;
    attntraffic) cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
            run attntraffic rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/atr -- python tools/bench_attn.py ${ATTN_ARGS:-}
            find gpurun_out/atr -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} gpurun_out/attn_fetch.csv
            rm -rf gpurun_out/atr ;;
    w8prof) cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
            run w8prof rocprofv3 --kernel-trace --output-format csv -d gpurun_out/w8prof -- python tools/bench_w8a8.py ${W8_ARGS:-}
            find gpurun_out/w8prof -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} gpurun_out/w8a8_trace.csv
            rm -rf gpurun_out/w8prof ;;
    gemmsweep) run gemmsweep python tools/bench_gemm.py --sweep ;;
    prof)   cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
            run prof rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 16 --warmup 4 --no-sweep --no-cpu-baseline ${BENCH_ARGS:-}
            find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats.csv
            head -40 gpurun_out/kernel_stats.csv ;;
    *)      echo "unknown step $step" ;;
  esac
done
