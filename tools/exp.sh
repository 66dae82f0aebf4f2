mkdir -p gpurun_out
{
echo "== default"; python tools/bench_gemm.py --ms 1,8,16
for wk in 1 2 4; do echo "== tall MT1 WK=$wk"; NMV_W4_TALL_MIN_M=1 NMV_W4_TALL_WK=$wk python tools/bench_gemm.py --ms 1,8,16; done
} > gpurun_out/exp1.log 2>&1
