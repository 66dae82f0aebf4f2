"""hipGraph-timed microbenchmark of the generic weight-only formats (csrc/wq_generic.hip): GPTQ
(exllama layout), AWQ and 8-bit Marlin, random codes, group 128.
usage: python tools/bench_wq.py [--ms 1,64] [--shapes gate_up,o]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neural_magic_vllm_amd import _custom_ops as ops  # noqa: E402

SHAPES = {"qkv": (4096, 6144), "o": (4096, 4096), "gate_up": (4096, 28672), "down": (14336, 4096)}


def timed(fn, iters=20):
    fn(0)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn(0)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters):
            fn(i)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", default="1,64")
    ap.add_argument("--shapes", default="o,gate_up")
    ap.add_argument("--generic", action="store_true", help="also the variants served by the generic kernel")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    gs = 128
    gen = torch.Generator(device=dev).manual_seed(0)

    def ri(shape):
        return torch.randint(-2**31, 2**31 - 1, shape, dtype=torch.int32, device=dev, generator=gen)

    for name in args.shapes.split(","):
        k, n = SHAPES[name]
        ncopy = max(2, (600 << 20) // (k * n // 2))
        sc = (torch.rand((k // gs, n), device=dev, generator=gen) * 0.01).half()
        g_idx = (torch.arange(k, device=dev) // gs).to(torch.int32)
        gq = [ri((k // 8, n)) for _ in range(ncopy)]
        gz = ri((k // gs, n // 8))
        aq = [ri((k, n // 8)) for _ in range(ncopy)]
        m8 = [ri((k // 16, n * 4)) for _ in range(max(2, ncopy // 2))]
        m4 = [ri((k // 16, n * 2)) for _ in range(ncopy)]
        zp = torch.randint(0, 16, (k // gs, n), device=dev, generator=gen).half()
        wsp = torch.zeros(n // 64 * 16, dtype=torch.int32, device=dev)
        e = torch.empty(0, dtype=torch.int32, device=dev)
        for m in [int(x) for x in args.ms.split(",")]:
            a = torch.randn((m, k), device=dev, dtype=torch.half)
            t_gptq = timed(lambda i: ops.gptq_gemm(a, gq[i % ncopy], gz, sc, e, True, 4))
            t_awq = timed(lambda i: ops.awq_gemm(a, aq[i % ncopy], sc, gz, 8))
            t_m8 = timed(lambda i: ops.gptq_marlin_gemm(a, m8[i % len(m8)], sc, e, e, wsp, 8, m, n, k, True))
            t_zp = timed(lambda i: ops.marlin_zp_gemm(a, m4[i % ncopy], sc, zp, wsp, m, n, k))
            b4, b8 = k * n // 2, k * n
            if args.generic:
                # the variants served by the generic kernel (wq_gemm_kernel)
                sc64 = (torch.rand((k // 64, n), device=dev, generator=gen) * 0.01).half()
                t_m8g = timed(lambda i: ops.gptq_marlin_gemm(a, m8[i % len(m8)], sc64, e, e, wsp, 8, m, n, k, True))
                # act-order on a K shard: sorted group runs of irregular length, scales of 4x as many groups
                runs = torch.sort(torch.randint(0, 4 * k // gs, (k, ), device=dev, generator=gen)).values.to(torch.int32)
                perm = torch.randperm(k, device=dev, generator=gen).to(torch.int32)
                sc_full = (torch.rand((4 * k // gs, n), device=dev, generator=gen) * 0.01).half()
                t_act = timed(lambda i: ops.gptq_marlin_gemm(a, m4[i % ncopy], sc_full, runs, perm, wsp, 4, m, n, k, False))
                sc1 = (torch.rand((1, n), device=dev, generator=gen) * 0.01).half()
                t_f8 = timed(lambda i: ops.fp8_marlin_gemm(a, m8[i % len(m8)], sc1, wsp, 8, m, n, k))
                g8 = [ri((k // 4, n)) for _ in range(2)]
                gz8 = ri((k // gs, n // 4))
                t_g8 = timed(lambda i: ops.gptq_gemm(a, g8[i % 2], gz8, sc, e, True, 8))
                g3 = [ri((k // 32 * 3, n)) for _ in range(2)]
                gz3 = ri((k // gs, n * 3 // 32))
                t_g3 = timed(lambda i: ops.gptq_gemm(a, g3[i % 2], gz3, sc, e, True, 3))
                print(f"{name:8s} M={m:3d}  generic: marlin8 g64 {t_m8g:7.1f} us {b8 / t_m8g / 1e3:5.0f} GB/s   "
                      f"marlin4 act-order K-shard {t_act:7.1f} us {b4 / t_act / 1e3:5.0f} GB/s   fp8-marlin {t_f8:7.1f} us "
                      f"{b8 / t_f8 / 1e3:5.0f} GB/s   gptq8 {t_g8:7.1f} us {b8 / t_g8 / 1e3:5.0f} GB/s   gptq3 {t_g3:7.1f} us "
                      f"{k * n * 3 // 8 / t_g3 / 1e3:5.0f} GB/s", flush=True)
            print(f"{name:8s} M={m:3d}  gptq4 {t_gptq:8.1f} us {b4 / t_gptq / 1e3:6.0f} GB/s   awq4 {t_awq:8.1f} us "
                  f"{b4 / t_awq / 1e3:6.0f} GB/s   marlin8 {t_m8:8.1f} us {b8 / t_m8 / 1e3:6.0f} GB/s   "
                  f"marlin4+zp {t_zp:8.1f} us {b4 / t_zp / 1e3:6.0f} GB/s", flush=True)
