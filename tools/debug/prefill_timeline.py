"""Phase times of the prefill kernel's main loop from its in-kernel timers (library built with -DNMV_W4P_STAMPS:
tools/debug/abl_src.sh w4a16_prefill pf_stamps=-DNMV_W4P_STAMPS, run with NMV_HIP_LIB=build/abl/libnmv_pf_stamps.so):
per wave role the mean shader clocks per stage in DMA issue / multiply / expansion / vmcnt wait / barrier.
usage: prefill_timeline.py [K N M [mode]]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neural_magic_vllm_amd import _custom_ops as ops, _lib  # noqa: E402

if __name__ == "__main__":
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
    lib = _lib.load()
    nwg = min(256, (n + 255) // 256)
    buf = (ctypes.c_ulonglong * (256 * 8 * 16))()
    lib.w4p_dbg_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert lib.w4p_dbg_stamps(buf, 256 * 8 * 16) == 0
    raw = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8, 16)[:nwg].astype(np.float64)
    stages = k // 32
    loop_us = raw[:, :, 8].mean() / 100.0
    print(f"K={k} N={n} M={m} mode={mode}: {nwg} workgroups of row block 0, {stages} stages; loop {loop_us:.1f} us "
          f"= {loop_us / stages * 1e3:.0f} ns per stage; shader clock ~ {raw[:, :, 7].mean() / loop_us / 1e3:.2f} GHz")
    names = ["DMA issue", "multiply", "expansion", "vmcnt wait", "barrier"]
    for role, sl in (("waves 0-3 (codes, expansion)", slice(0, 4)), ("waves 4-7 (activations)", slice(4, 8))):
        tot = raw[:, sl, 7].mean() / stages
        print(f"  {role}: {tot:.0f} clocks per stage")
        for i, nm in enumerate(names):
            v = raw[:, sl, i] / stages
            print(f"    {nm:11s}: mean {v.mean():7.0f}  (min {v.min():7.0f}, max {v.max():7.0f})")
