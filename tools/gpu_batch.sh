#!/bin/bash
# scratch driver of one gpurun call (edited per call; see tools/gpu_ci.sh for the standing steps)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_cache.py tests/test_gpu_attention.py tests/test_gpu_linear_methods.py tests/test_gpu_checkpoints.py tests/test_gpu_model.py tests/test_gpu_custom_ar_ops.py tests/test_gpu_custom_allreduce.py tests/test_gpu_tp.py -q -m gpu --timeout 600 -x > gpurun_out/r3_t_decopy.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r3_t_decopy.log
for v in "" "NMV_W4_SPLITS=2" "NMV_W4_SPLITS=3" "NMV_W4_TALL_MT=4" "NMV_W4_TALL_MT=4 NMV_W4_SPLITS=2"; do
  echo "== $v"; env $v timeout -k 10 120 python tools/bench_gemm.py --ms 512 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r3_gemm512_variants.log 2>&1; cat gpurun_out/r3_gemm512_variants.log
