"""hipGraph-timed microbenchmark of the glue ops at Llama-3-8B sizes (per-call device time incl.
the launch boundary inside a graph).  usage: python tools/bench_glue.py [--batches 1,64]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neural_magic_vllm_amd import _custom_ops as ops  # noqa: E402


def timed(fn, iters=50):
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
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
    ap.add_argument("--batches", default="1,8,64")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    H, I, NQ, NKV, D = 4096, 14336, 32, 8, 128
    for b in [int(x) for x in args.batches.split(",")]:
        x = torch.randn(b, H, device=dev, dtype=torch.bfloat16)
        res = torch.randn(b, H, device=dev, dtype=torch.bfloat16)
        w = torch.ones(H, device=dev, dtype=torch.bfloat16)
        out = torch.empty_like(x)
        gu = torch.randn(b, 2 * I, device=dev, dtype=torch.bfloat16)
        act = torch.empty(b, I, device=dev, dtype=torch.bfloat16)
        qkv = torch.randn(b, (NQ + 2 * NKV) * D, device=dev, dtype=torch.bfloat16)
        q, k, v = qkv.split([NQ * D, NKV * D, NKV * D], dim=-1)
        pos = torch.full((b, ), 512, dtype=torch.int64, device=dev)
        cs = torch.randn(8192, D, device=dev, dtype=torch.bfloat16)
        nb = 4096
        kc = torch.empty(nb, NKV, D // 8, 16, 8, device=dev, dtype=torch.bfloat16)
        vc = torch.empty(nb, NKV, D, 16, device=dev, dtype=torch.bfloat16)
        slots = (torch.randperm(nb, device=dev)[:b] * 16 + 3).to(torch.int64)
        r = {
            "rms_norm": timed(lambda: ops.rms_norm(out, x, w, 1e-5)),
            "fused_add_rms_norm": timed(lambda: ops.fused_add_rms_norm(x, res, w, 1e-5)),
            "rotary_embedding": timed(lambda: ops.rotary_embedding(pos, q, k, D, cs, True)),
            "reshape_and_cache": timed(lambda: ops.reshape_and_cache(k.view(b, NKV, D), v.view(b, NKV, D), kc, vc, slots, "auto", 1.0)),
            "silu_and_mul": timed(lambda: ops.silu_and_mul(act, gu)),
        }
        print(f"B={b:3d}  " + "  ".join(f"{k_}={v_:.2f}us" for k_, v_ in r.items()), flush=True)
