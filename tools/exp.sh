mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for i in 1 2; do
  case $i in
   1) PMC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS" ;;
   2) PMC="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC" ;;
  esac
  timeout -k 10 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d gpurun_out/gpmc$i -- python tools/bench_gemm.py --ms 1,64 --shapes gate_up,down > gpurun_out/pmc$i.log 2>&1 || { echo "pass $i failed"; tail -5 gpurun_out/pmc$i.log; }
  find gpurun_out/gpmc$i -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} gpurun_out/gemm_pmc$i.csv
  rm -rf gpurun_out/gpmc$i
done
