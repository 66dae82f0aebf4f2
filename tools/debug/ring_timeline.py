"""Timeline of one ring-kernel launch from its in-kernel stamps (library built with -DNMV_W4R_DEBUG; NMV_W4R_DBG bit 32):
per stamp point the min / median / max over waves of the time since the earliest entry, consumers and loaders apart.
points: 0 entry, 1 flags zeroed + barrier, 2 first group landed (loader) / seen (consumer), 3 half of the groups, 4 loop end,
5 row sums published (loader) / zero-point done (consumer), 6 after the barrier, 7 k-lane image written + barrier."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neural_magic_vllm_amd import _custom_ops as ops, _lib  # noqa: E402

if __name__ == "__main__":
    k, n, m = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (4096, 28672, 64)))
    mode = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    extra = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    dev = torch.device("cuda:0")
    os.environ["NMV_W4R"] = "1"
    g = torch.Generator(device=dev).manual_seed(0)
    ws = [torch.randint(-2**31, 2**31 - 1, (k // 8 * n,), dtype=torch.int32, device=dev, generator=g) for _ in range(6)]
    sc = (torch.rand((k // 128, n), device=dev, generator=g) * 0.01).to(torch.bfloat16)
    a = torch.randn((m, k), device=dev, dtype=torch.bfloat16)
    wsp = torch.zeros(n // 64 * 16, dtype=torch.int32, device=dev)
    os.environ["NMV_W4R_DBG"] = str(extra)
    for i in range(5):
        ops.w4_native_gemm(a, ws[i], sc, wsp, m, n, k, mode)
    torch.cuda.synchronize()
    os.environ["NMV_W4R_DBG"] = str(32 | extra)
    ops.w4_native_gemm(a, ws[5], sc, wsp, m, n, k, mode)
    torch.cuda.synchronize()
    lib = _lib.load()
    nwg = min(256, (n // 64 + 1) // 2)
    buf = (ctypes.c_ulonglong * (256 * 16 * 16))()
    lib.w4r_dbg_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert lib.w4r_dbg_stamps(buf, 256 * 16 * 16) == 0
    raw = np.frombuffer(buf, dtype=np.uint64).reshape(256, 16, 16)[:nwg].astype(np.float64)
    st = raw[:, :, :8]
    t0 = st[:, :, 0].min()
    us = (st - t0) / 100.0
    print(f"K={k} N={n} M={m} mode={mode} dbg={extra}: {nwg} workgroups, 100 MHz stamps, us since the first wave entered")
    for role, sl in (("consumer", slice(0, 12)), ("loader", slice(12, 16))):
        for p in range(8):
            v = us[:, sl, p].ravel()
            print(f"  {role:8s} point {p}: min {v.min():6.2f}  med {np.median(v):6.2f}  max {v.max():6.2f}")
    # summed phase times in shader clocks (s_memtime): consumers 0 = compute + reads, 1 = waiting for FULL;
    # loaders 0 = waiting for FREE, 1 = DMA issue, 2 = vmcnt wait, 3 = publish + row sums
    loop_us = np.median(us[:, 0:8, 4] - us[:, 0:8, 2])
    tot = raw[:, 0:8, 8].mean() + raw[:, 0:8, 9].mean()
    print(f"  shader clock ~ {tot / loop_us / 1e3:.2f} GHz (consumer loop {loop_us:.2f} us = {tot:.0f} clocks)")
    names = {"consumer": ["compute+reads", "wait FULL"], "act loader": ["wait FREE", "DMA issue", "vmcnt wait", "publish"], "code loader": ["wait FREE", "DMA issue", "vmcnt wait", "publish"]}
    for role, sl in (("consumer", slice(0, 8)), ("act loader", slice(8, 10)), ("code loader", slice(10, 12))):
        for k, nm in enumerate(names[role]):
            v = raw[:, sl, 8 + k].ravel()
            print(f"  {role:11s} {nm:14s}: mean {v.mean():9.0f} clocks  (min {v.min():9.0f}, max {v.max():9.0f})")
