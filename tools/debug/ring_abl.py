"""Ablations of the ring kernel: times every variant library under build/abl/ (tools/debug/abl_w4r.sh name=-DNMV_W4R_ABL=bits:\n1 no row sums, 2 no MFMA, 4 no operand reads, 8 no expansion / MFMA, 16 no DMA; results garbage, times valid)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_gemm import SHAPES, bench  # noqa: E402

if __name__ == "__main__":
    # one process per variant library: NMV_HIP_LIB is read at import
    import subprocess
    shapes = sys.argv[1] if len(sys.argv) > 1 else "gate_up"
    if os.environ.get("RING_ABL_CHILD"):
        dev = torch.device("cuda:0")
        os.environ["NMV_W4R"] = "1"
        out = []
        for name in shapes.split(","):
            k, n = SHAPES[name]
            md = 1 if name.startswith("gate_up") else 2
            us, _ = bench(name, k, n, 64, dev, iters=16, native=md)
            out.append(f"{name}:{us:.1f}")
        print(os.environ["RING_ABL_CHILD"], " ".join(out), flush=True)
        sys.exit(0)
    import glob
    libs = [("product", "")] + [(os.path.basename(f)[7:-3], f) for f in sorted(glob.glob(os.path.join(ROOT, "build/abl/libnmv_*.so")))]
    for tag, lib in libs:
        env = dict(os.environ, RING_ABL_CHILD=tag)
        if lib:
            env["NMV_HIP_LIB"] = lib
        subprocess.run([sys.executable, os.path.abspath(__file__), shapes], env=env, check=False)
