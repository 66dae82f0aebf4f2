"""hipGraph-timed microbenchmark of cutlass_scaled_mm (W8A8 int8 / fp8) on the Llama-3-8B layer
shapes; weights rotated through > 600 MB so they come from HBM.
usage: python tools/bench_w8a8.py [--ms 1,16,64] [--dtype int8|fp8]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neural_magic_vllm_amd import _custom_ops as ops  # noqa: E402

SHAPES = {"qkv": (4096, 6144), "o": (4096, 4096), "gate_up": (4096, 28672), "down": (14336, 4096)}


def bench(k, n, m, dev, fp8, iters=40):
    ncopy = max(2, (600 << 20) // (k * n))
    g = torch.Generator(device=dev).manual_seed(0)
    if fp8:
        bs = [torch.randint(0, 120, (n, k), dtype=torch.uint8, device=dev, generator=g).view(torch.float8_e4m3fn).t()
              for _ in range(ncopy)]
        a = torch.randint(0, 120, (m, k), dtype=torch.uint8, device=dev, generator=g).view(torch.float8_e4m3fn)
    else:
        bs = [torch.randint(-127, 127, (n, k), dtype=torch.int8, device=dev, generator=g).t() for _ in range(ncopy)]
        a = torch.randint(-127, 127, (m, k), dtype=torch.int8, device=dev, generator=g)
    sa = torch.rand((m, 1), device=dev, generator=g) * 0.01
    sb = torch.rand((1, n), device=dev, generator=g) * 0.01

    def run(i):
        return ops.cutlass_scaled_mm(a, bs[i % ncopy], sa, sb, torch.bfloat16)

    run(0)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run(0)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for i in range(iters):
            run(i)
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    graph.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    alg = k * n + m * k + 2 * m * n + 4 * (m + n)
    return us, alg / us / 1e3


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", default="1,16,64")
    ap.add_argument("--dtype", default="int8")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for name, (k, n) in SHAPES.items():
        for m in [int(x) for x in args.ms.split(",")]:
            us, gbs = bench(k, n, m, dev, args.dtype == "fp8")
            print(f"{args.dtype} {name:8s} M={m:3d}  {us:8.1f} us  {gbs:7.0f} GB/s", flush=True)
