#!/bin/bash
# SQ / LDS / traffic counters of the prefill kernel (one rocprofv3 --pmc pass per counter set; summaries via tools/pmc_summary.py)
set -o pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
ARGS=${ARGS:-4096 28672 2048 1}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT" \
           "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1)); rm -rf /tmp/_pp
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/_pp -- python3 tools/debug/prefill_pmc_run.py $ARGS > $OUT/prefill_pmc_$i.log 2>&1 \
    || { echo "set $i failed"; tail -3 $OUT/prefill_pmc_$i.log; continue; }
  f=$(find /tmp/_pp -name "*counter_collection.csv" | head -1)
  python3 tools/pmc_summary.py $f "prefill_kernel"
done
