"""a few launches of the prefill kernel for rocprofv3 --pmc (tools/debug/prefill_pmc.sh).  usage: prefill_pmc_run.py [K N M mode]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neural_magic_vllm_amd import _custom_ops as ops  # noqa: E402

k, n, m = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (4096, 28672, 2048)))
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device("cuda:0")
os.environ["NMV_W4P"] = "2"
os.environ["NMV_W4P_SPLITS"] = "1"
g = torch.Generator(device=dev).manual_seed(0)
ws = [torch.randint(-2**31, 2**31 - 1, (k // 8 * n,), dtype=torch.int32, device=dev, generator=g) for _ in range(4)]
sc = (torch.rand((k // 128, n), device=dev, generator=g) * 0.01).to(torch.bfloat16)
a = torch.randn((m, k), device=dev, dtype=torch.bfloat16)
wsp = torch.zeros(n // 64 * 16, dtype=torch.int32, device=dev)
for i in range(4):
    ops.w4_native_gemm(a, ws[i], sc, wsp, m, n, k, mode)
torch.cuda.synchronize()
