#!/bin/bash
# scratch driver of one gpurun call (edited per call; see tools/gpu_ci.sh for the standing steps)
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_attention.py -q -m gpu -k "rope_partial" --timeout 200 > gpurun_out/t_attn_rp.log 2>&1; echo "attn rc=$?"; tail -3 gpurun_out/t_attn_rp.log
timeout -k 10 400 python -m pytest tests/test_gpu_wq_formats.py -q -m gpu --timeout 200 > gpurun_out/t_wq.log 2>&1; echo "wq rc=$?"; tail -3 gpurun_out/t_wq.log
timeout -k 10 200 python tools/bench_wq.py --generic --ms 1,64 2>&1 | grep -v amdgpu.ids > gpurun_out/r3_wq.log; cat gpurun_out/r3_wq.log
echo "== failed-capture rehearsal"
NMV_BENCH_DIST_BACKEND=gloo NMV_BENCH_SINGLE_DEVICE=1 NMV_CUSTOM_ALLREDUCE=force NMV_CUSTOM_AR_TIMEOUT_MS=30000 NMV_TEST_FAIL_CAPTURE_RANK=1 \
  timeout -k 10 240 python bench.py --gpus 2 --steps 4 --warmup 2 --model tiny --batch 4 --context 40 --no-sweep > gpurun_out/fc.out 2> gpurun_out/fc.err
echo "rc=$?"; tail -c 600 gpurun_out/fc.out; tail -25 gpurun_out/fc.err
