set -e
python -m pytest tests/test_gpu_w4a16.py -x -q -m gpu -k "tall" 2>&1 | tail -3
echo "== default"; python tools/bench_gemm.py --ms 64,128,256 2>/dev/null | grep -v "^#"
for wk in 1 2 4; do echo "== MT=4 occ1 wk=$wk"; NMV_W4_TALL_MT=4 NMV_W4_TALL_WK=$wk python tools/bench_gemm.py --ms 64,128,256 2>/dev/null | grep -v "^#"; done
