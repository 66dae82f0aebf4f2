#!/bin/bash
# Round-4 profiles on the gpurun box: kernel trace, HBM traffic (FETCH_SIZE / WRITE_SIZE, one counter per pass) and SQ counters
# of the four W4A16 launches of a decoder layer as the decode step issues them (tools/bench_step_gemms.py), and the kernel
# stats of the whole decode step (bench.py) at B = 64 and B = 1.  Raw CSVs stay in gpurun_out/; summaries go to profiles/.
set -o pipefail
REPO=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
MS=${MS:-32,64}; NGR=${NGR:-24}   # (GROUPS is a bash builtin array)
one() {  # name, rocprofv3 args..., then -- program
  name=$1; shift
  rm -rf gpurun_out/_p_$name
  timeout -k 10 400 rocprofv3 "$@" > gpurun_out/r04_$name.log 2>&1 || { echo "rocprofv3 $name failed"; tail -5 gpurun_out/r04_$name.log; return 1; }
}
one trace --kernel-trace --output-format csv -d gpurun_out/_p_trace -- python3 tools/bench_step_gemms.py --ms $MS --groups $NGR &&
  find gpurun_out/_p_trace -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} gpurun_out/r04_step_trace.csv
for c in FETCH_SIZE WRITE_SIZE; do
  one $c --pmc $c --kernel-trace --output-format csv -d gpurun_out/_p_$c -- python3 tools/bench_step_gemms.py --ms $MS --groups $NGR &&
    find gpurun_out/_p_$c -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} gpurun_out/r04_step_$c.csv
done
one sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU \
    --kernel-trace --output-format csv -d gpurun_out/_p_sq -- python3 tools/bench_step_gemms.py --ms $MS --groups $NGR &&
  find gpurun_out/_p_sq -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} gpurun_out/r04_step_sq.csv
one sq2 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_SALU SQ_INSTS_VMEM \
    --kernel-trace --output-format csv -d gpurun_out/_p_sq2 -- python3 tools/bench_step_gemms.py --ms $MS --groups $NGR &&
  find gpurun_out/_p_sq2 -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} gpurun_out/r04_step_sq2.csv
for b in 64 1; do
  one bench_b$b --kernel-trace --stats --output-format csv -d gpurun_out/_p_bench$b -- python3 bench.py --steps 16 --warmup 4 --no-sweep --no-cpu-baseline --batch $b &&
    find gpurun_out/_p_bench$b -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r04_bench_kernel_stats_b$b.csv
done
rm -rf gpurun_out/_p_*
python3 tools/step_gemm_summary.py --ms $MS --groups $NGR --trace gpurun_out/r04_step_trace.csv --fetch gpurun_out/r04_step_FETCH_SIZE.csv \
    --write gpurun_out/r04_step_WRITE_SIZE.csv --json gpurun_out/r04_step_gemm_traffic.json | tee gpurun_out/r04_step_gemm_summary.txt
python3 tools/pmc_summary.py gpurun_out/r04_step_sq.csv "w4a16" > gpurun_out/r04_step_pmc.txt
python3 tools/pmc_summary.py gpurun_out/r04_step_sq2.csv "w4a16" >> gpurun_out/r04_step_pmc.txt
tail -30 gpurun_out/r04_step_pmc.txt
rm -f gpurun_out/r04_step_trace.csv gpurun_out/r04_step_FETCH_SIZE.csv gpurun_out/r04_step_WRITE_SIZE.csv gpurun_out/r04_step_sq.csv gpurun_out/r04_step_sq2.csv
ls -la gpurun_out/r04_* | head -30
