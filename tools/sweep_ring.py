"""GPU: the LDS-DMA ring kernel (NMV_W4_RING=1) against the tall kernel on the Llama-3-8B projections."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_gemm  # noqa: E402

KNOBS = ("NMV_W4_RING", "NMV_W4_DIRECT", "NMV_W4_DIRECT_WK", "NMV_W4_SPLITS")


def clear():
    for k in KNOBS:
        os.environ.pop(k, None)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", default="1,16,32")
    ap.add_argument("--shapes", default=bench_gemm.DEFAULT_SHAPES)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for name in args.shapes.split(","):
        k, n = bench_gemm.SHAPES[name]
        for m in [int(x) for x in args.ms.split(",")]:
            clear()
            base, gbs = bench_gemm.bench(name, k, n, m, dev, iters=24)
            res = []
            for wk in (4, 2, 1):
                for sp in (1, 2, 4, 7, 8, 14):
                    clear()
                    os.environ.update(NMV_W4_RING="1", NMV_W4_DIRECT_WK=str(wk), NMV_W4_SPLITS=str(sp))
                    try:
                        us, _ = bench_gemm.bench(name, k, n, m, dev, iters=24)
                    except Exception:
                        continue
                    res.append((us, wk, sp))
            clear()
            res.sort()
            best = ", ".join(f"wk{w}/sp{s}:{u:.1f}" for u, w, s in res[:6])
            print(f"{name:8s} M={m:3d} tall {base:6.1f} us ({gbs:5.0f} GB/s) | ring best {best}", flush=True)
