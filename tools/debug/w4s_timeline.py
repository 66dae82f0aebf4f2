"""Phase timeline of w4a16_stream_kernel from a diagnostic build:
   make -C neural_magic_vllm_amd/csrc BUILD=$PWD/build/hip_stamps OUT=$PWD/build/libnmvllm_hip_stamps.so EXTRA=-DNMV_W4S_STAMPS
   NMV_HIP_LIB=build/libnmvllm_hip_stamps.so python tools/debug/w4s_timeline.py --shape gate_up --m 1
Stamps (100 MHz wall clock per wave): 0 entry, 1 ring filled (loads issued), 2 activations / scales staged, 3 past the
barrier, 4 end of the main loop, 5 end of the k-group reduction.  Times are relative to the first wave's entry."""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from neural_magic_vllm_amd import _custom_ops as ops, _lib  # noqa: E402
from bench_gemm import SHAPES  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="gate_up")
ap.add_argument("--m", type=int, default=1)
ap.add_argument("--waves", type=int, default=8)
args = ap.parse_args()
dev = torch.device("cuda:0")
k, n = SHAPES[args.shape]
m = args.m
g = torch.Generator(device=dev).manual_seed(0)
ncopy = 12
ws = [torch.randint(-2**31, 2**31 - 1, (k // 16, n * 2), dtype=torch.int32, device=dev, generator=g) for _ in range(ncopy)]
sc = (torch.rand((k // 128, n), device=dev, generator=g) * 0.01).to(torch.bfloat16)
a = torch.randn((m, k), device=dev, dtype=torch.bfloat16)
wsp = torch.zeros(n // 64 * 16, dtype=torch.int32, device=dev)
e = torch.empty(0, dtype=torch.int32, device=dev)
for i in range(ncopy):
    ops.gptq_marlin_gemm(a, ws[i], sc, e, e, wsp, 4, m, n, k, True)
torch.cuda.synchronize()
lib = _lib.load()
lib.nmv_dbg_w4s_stamps.restype = ctypes.c_int
nst = 1 << 18
host = (ctypes.c_ulonglong * nst)()
# clear by reading, then one more call on a cold copy
ops.gptq_marlin_gemm(a, ws[0], sc, e, e, wsp, 4, m, n, k, True)
torch.cuda.synchronize()
assert lib.nmv_dbg_w4s_stamps(host, nst) == 0
st = np.frombuffer(host, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
st = st[st[:, 0] > 0]
t0 = st[:, 0].min()
rel = (st[:, :6] - t0) * 0.01   # us
names = ["entry", "ring issued", "staged", "past barrier", "loop end", "reduced"]
print(f"{args.shape} M={m}: {len(st)} waves")
for i, nm in enumerate(names):
    c = rel[:, i]
    print(f"  {nm:14s} mean {c.mean():6.2f}  min {c.min():6.2f}  p50 {np.median(c):6.2f}  p90 {np.percentile(c, 90):6.2f}  max {c.max():6.2f} us")
d = rel[:, 4] - rel[:, 3]
print(f"  main loop duration per wave: mean {d.mean():.2f}  min {d.min():.2f}  max {d.max():.2f} us")
if st[:, 7].max() > 0:   # streamed kernels: per-stage phases summed over the stages (100 MHz ticks)
    comp = (st[:, 6] & 0xffffffff) * 0.01
    park = (st[:, 6] >> 32) * 0.01
    bar = st[:, 7] * 0.01
    print(f"  in-loop, per wave: compute {comp.mean():.2f}  park (wait for the stage's loads + LDS stores) {park.mean():.2f}  "
          f"barrier {bar.mean():.2f} us")
