# gate_up / down / qkv at M = 64: two co-resident 32-row workgroups (128 registers, 4 waves per SIMD) instead of the 64-row tile
env NMV_W4S_MT=2 NMV_W4S_GST=1 NMV_W4S_D=1 NMV_W4S_STRICT=1 python tools/bench_gemm.py --native --ms 33,64 --shapes gate_up,down,qkv
