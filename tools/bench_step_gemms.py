"""The four W4A16 launches of one Llama-3-8B decoder layer exactly as the decode step issues them -- nmv_w4_native_gemm on
the MFMA-native tensor: qkv / o / down in mode 2 (deferred split-K: fp32 slabs left to the next launch), gate_up in mode 1
(silu(gate) * up in the epilogue) -- `--groups` launch groups per M, weights rotated through > 600 MB.  Meant to run under
rocprofv3 (kernel trace or one PMC counter per pass): tools/step_gemm_summary.py turns the CSVs into per-projection
time / HBM bytes.  usage: python tools/bench_step_gemms.py [--ms 32,64] [--groups 24]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neural_magic_vllm_amd import _custom_ops as ops  # noqa: E402

SHAPES = [("qkv", 4096, 6144, 2), ("o", 4096, 4096, 2), ("gate_up", 4096, 28672, 1), ("down", 14336, 4096, 2)]

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", default="32,64")
    ap.add_argument("--groups", type=int, default=24)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    ncopy = 6
    W = {name: [torch.randint(-2**31, 2**31 - 1, (k // 8 * n,), dtype=torch.int32, device=dev, generator=g) for _ in range(ncopy)]
         for name, k, n, _ in SHAPES}
    S = {name: (torch.rand((k // 128, n), device=dev, generator=g) * 0.01).to(torch.bfloat16) for name, k, n, _ in SHAPES}
    wsp = torch.zeros(28672 // 64 * 16, dtype=torch.int32, device=dev)
    for m in [int(x) for x in args.ms.split(",")]:
        A = {name: torch.randn((m, k), device=dev, dtype=torch.bfloat16) for name, k, n, _ in SHAPES}
        for i in range(args.groups):
            for name, k, n, mode in SHAPES:
                ops.w4_native_gemm(A[name], W[name][i % ncopy], S[name], wsp, m, n, k, mode)
        torch.cuda.synchronize()
        print(f"M={m}: {args.groups} launch groups issued", flush=True)
