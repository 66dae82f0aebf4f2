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

SHAPES = {"qkv": (4096, 6144), "o": (4096, 4096), "gate_up": (4096, 28672), "down": (14336, 4096),
          # Llama-3-70B, TP=8, per rank (BASELINE.json configs[4])
          "qkv70": (8192, 1280), "o70": (1024, 8192), "gate_up70": (8192, 7168), "down70": (3584, 8192),
          # Llama-3-8B gate_up under TP = 2 / 4 / 8 (224 / 112 / 56 chunks: where the fused silu form starts to pay)
          "gate_up_tp2": (4096, 14336), "gate_up_tp4": (4096, 7168), "gate_up_tp8": (4096, 3584),
          # the other Llama-3-8B projections under TP = 4 / 8
          "qkv_tp4": (4096, 1536), "qkv_tp8": (4096, 768), "o_tp4": (1024, 4096), "o_tp8": (512, 4096),
          "down_tp4": (3584, 4096), "down_tp8": (1792, 4096),
          "qkv_tp2": (4096, 3072), "o_tp2": (2048, 4096), "down_tp2": (7168, 4096)}
DEFAULT_SHAPES = "qkv,o,gate_up,down"


def bench(name, k, n, m, dev, iters=40, gs=128, native=None, mode=0):
    """native: None = the Marlin ops (mode 0 gptq_marlin_gemm, 1 ..._silu_mul, 2 ..._partial);
    0 / 1 / 2 = nmv_w4_native_gemm in that mode"""
    nbytes = k * n // 2
    # NMV_BENCH_NCOPY=1: the same weight tensor every call -- served by the 256 MiB Infinity Cache, not HBM
    ncopy = int(os.environ.get("NMV_BENCH_NCOPY", max(2, (600 << 20) // nbytes)))
    g = torch.Generator(device=dev).manual_seed(0)
    ws = [torch.randint(-2**31, 2**31 - 1, (k // 16, n * 2), dtype=torch.int32, device=dev, generator=g)
          for _ in range(ncopy)]
    if native is not None:
        ws = [w.view(-1) for w in ws]
    sc = [(torch.rand((k // gs, n), device=dev, generator=g) * 0.01).to(torch.bfloat16) for _ in range(ncopy)]
    a = torch.randn((m, k), device=dev, dtype=torch.bfloat16)
    wsp = torch.zeros(n // 64 * 16, dtype=torch.int32, device=dev)
    e = torch.empty(0, dtype=torch.int32, device=dev)
    def call(i):
        if native is None:
            if mode == 1:
                return ops.gptq_marlin_gemm_silu_mul(a, ws[i], sc[i], wsp, m, n, k)
            if mode == 2:
                return ops.gptq_marlin_gemm_partial(a, ws[i], sc[i], m, n, k)
            return ops.gptq_marlin_gemm(a, ws[i], sc[i], e, e, wsp, 4, m, n, k, True)
        return ops.w4_native_gemm(a, ws[i], sc[i], wsp, m, n, k, native)

    for i in range(min(ncopy, 3)):
        call(i)
    torch.cuda.synchronize()
    # capture `iters` back-to-back calls in one hipGraph: device time without host launch cost
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        call(0)
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(graph):
        for i in range(iters):
            call(i % ncopy)
    graph.replay()
    torch.cuda.synchronize()
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    graph.replay()
    e1.record(st)
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    alg = k * n // 2 + (k // gs) * n * 2 + 2 * m * k + 2 * m * n
    return us, alg / us / 1e3


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--ms", default="1,16,32,64")
    ap.add_argument("--gs", type=int, default=128, help="group size; -1 = channelwise")
    ap.add_argument("--shapes", default=DEFAULT_SHAPES)
    ap.add_argument("--native", action="store_true", help="also time nmv_w4_native_gemm (modes 0 and 2)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ms = [int(x) for x in args.ms.split(",")]
    for name in args.shapes.split(","):
        k, n = SHAPES[name]
        for m in ms:
            if not args.sweep:
                us, gbs = bench(name, k, n, m, dev, gs=k if args.gs == -1 else args.gs)
                extra = ""
                if args.native:
                    gs = k if args.gs == -1 else args.gs
                    n0, g0 = bench(name, k, n, m, dev, gs=gs, native=0)
                    extra = f"   native {n0:6.1f} us {g0:6.0f} GB/s"
                    # the fused forms the decode step issues: silu-mul epilogue on gate_up, deferred split-K elsewhere
                    md = 1 if name.startswith("gate_up") else 2
                    mm, _ = bench(name, k, n, m, dev, gs=gs, mode=md)
                    nm, _ = bench(name, k, n, m, dev, gs=gs, native=md)
                    extra += f"   {'silu-mul' if md == 1 else 'deferred'}: marlin {mm:6.1f} us  native {nm:6.1f} us"
                print(f"{name:8s} M={m:3d}  {us:8.1f} us  {gbs:7.0f} GB/s{extra}", flush=True)
                continue
            res = []
            for mt in (0, 1, 2, 4):
                if mt and mt * 16 > max(m, 16):
                    continue
                for wn in (1, 2, 4):
                    for sp in (1, 2, 4, 8, 16, 32):
                        for kk in ("NMV_W4_WN", "NMV_W4_SPLITS", "NMV_W4_MT"):
                            os.environ.pop(kk, None)
                        if mt:
                            os.environ["NMV_W4_MT"] = str(mt)
                            os.environ["NMV_W4_WN"] = str(wn)
                            os.environ["NMV_W4_SPLITS"] = str(sp)
                        elif wn != 1 or sp != 1:
                            continue
                        try:
                            us, gbs = bench(name, k, n, m, dev, iters=12)
                        except Exception:
                            continue
                        res.append((us, mt, wn, sp, gbs))
            for kk in ("NMV_W4_WN", "NMV_W4_SPLITS", "NMV_W4_MT"):
                os.environ.pop(kk, None)
            res.sort()
            best = ", ".join(f"mt{a}/wn{w}/sp{s}:{u:.1f}" for u, a, w, s, _ in res[:8])
            dflt = [r for r in res if r[1] == 0]
            print(f"{name:8s} M={m:3d} default {dflt[0][0]:.1f}us {dflt[0][4]:.0f}GB/s | best(us) {best}", flush=True)
