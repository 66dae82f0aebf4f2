"""Ablations of the prefill kernel: times every variant library build/abl/libnmv_pf*.so (tools/debug/abl_src.sh w4a16_prefill
name=-DNMV_W4P_ABL=bits: 1 no activation DMA, 2 no code DMA, 4 no expansion, 8 no operand reads, 16 no MFMA; results
garbage, times valid).  usage: prefill_abl.py [shapes] [M]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_gemm import SHAPES, bench  # noqa: E402

if __name__ == "__main__":
    import subprocess
    shapes = sys.argv[1] if len(sys.argv) > 1 else "gate_up"
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    if os.environ.get("PF_ABL_CHILD"):
        dev = torch.device("cuda:0")
        os.environ["NMV_W4P"] = "2"
        os.environ["NMV_W4P_SPLITS"] = "1"
        out = []
        for name in shapes.split(","):
            k, n = SHAPES[name]
            md = 1 if name.startswith("gate_up") else 0
            us, _ = bench(name, k, n, m, dev, iters=8, native=md)
            out.append(f"{name}:{us:.1f}")
        print(os.environ["PF_ABL_CHILD"], f"M={m}", " ".join(out), flush=True)
        sys.exit(0)
    import glob
    libs = [("product", "")] + [(os.path.basename(f)[7:-3], f) for f in sorted(glob.glob(os.path.join(ROOT, "build/abl/libnmv_pf*.so")))]
    for tag, lib in libs:
        env = dict(os.environ, PF_ABL_CHILD=tag)
        if lib:
            env["NMV_HIP_LIB"] = lib
        subprocess.run([sys.executable, os.path.abspath(__file__), shapes, str(m)], env=env, check=False)
