"""GPU microbenchmark of gptq_marlin_gemm on the Llama-3-8B layer shapes (HIP events on the launch
stream; weights rotated through > 512 MB of copies so they come from HBM, not the Infinity Cache).
usage: python tools/bench_gemm.py [--sweep]  -- prints us per call and GB/s (algorithmic bytes)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neural_magic_vllm_amd import _custom_ops as ops  # noqa: E402

SHAPES = {"qkv": (4096, 6144), "o": (4096, 4096), "gate_up": (4096, 28672), "down": (14336, 4096)}


def bench(name, k, n, m, dev, iters=40, gs=128):
    nbytes = k * n // 2
    ncopy = max(2, (600 << 20) // nbytes)
    g = torch.Generator(device=dev).manual_seed(0)
    ws = [torch.randint(-2**31, 2**31 - 1, (k // 16, n * 2), dtype=torch.int32, device=dev, generator=g)
          for _ in range(ncopy)]
    sc = [(torch.rand((k // gs, n), device=dev, generator=g) * 0.01).to(torch.bfloat16) for _ in range(ncopy)]
    a = torch.randn((m, k), device=dev, dtype=torch.bfloat16)
    wsp = torch.zeros(n // 64 * 16, dtype=torch.int32, device=dev)
    e = torch.empty(0, dtype=torch.int32, device=dev)
    for i in range(ncopy):
        ops.gptq_marlin_gemm(a, ws[i], sc[i], e, e, wsp, 4, m, n, k, True)
    torch.cuda.synchronize()
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for i in range(iters):
        ops.gptq_marlin_gemm(a, ws[i % ncopy], sc[i % ncopy], e, e, wsp, 4, m, n, k, True)
    e1.record(st)
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    alg = k * n // 2 + (k // gs) * n * 2 + 2 * m * k + 2 * m * n
    return us, alg / us / 1e3


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--ms", default="1,16,64")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ms = [int(x) for x in args.ms.split(",")]
    for name, (k, n) in SHAPES.items():
        for m in ms:
            if not args.sweep:
                us, gbs = bench(name, k, n, m, dev)
                print(f"{name:8s} M={m:3d}  {us:8.1f} us  {gbs:7.0f} GB/s", flush=True)
                continue
            res = []
            for wn in (0, 1, 2, 4):
                for sp in (0, 1, 2, 4, 8, 16, 32):
                    os.environ.pop("NMV_W4_WN", None)
                    os.environ.pop("NMV_W4_SPLITS", None)
                    if wn:
                        os.environ["NMV_W4_WN"] = str(wn)
                    if sp:
                        os.environ["NMV_W4_SPLITS"] = str(sp)
                    if m > 32 and wn == 4:
                        continue
                    try:
                        us, gbs = bench(name, k, n, m, dev, iters=20)
                    except Exception as ex:
                        continue
                    res.append((us, wn, sp, gbs))
            res.sort()
            best = ", ".join(f"wn{w}/sp{s}:{u:.1f}us" for u, w, s, _ in res[:6])
            dflt = [r for r in res if r[1] == 0 and r[2] == 0]
            print(f"{name:8s} M={m:3d} default {dflt[0][0]:.1f}us {dflt[0][3]:.0f}GB/s | best {best}", flush=True)
