"""cutlass_scaled_mm at 17..64 rows: the K-splitting kernel (NMV_MM_WIDE=0) against the wide kernel (waves split N,
activations through LDS) with the plan's choice and with forced column-tile counts / slice counts.
usage: python tools/sweep_w8a8_wide.py [--ms 32,64] [--dtype int8]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_w8a8 import SHAPES, bench  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", default="32,64")
    ap.add_argument("--dtype", default="int8")
    ap.add_argument("--forms", default="4:1,4:2,2:1,2:2,2:4,1:2,1:4,1:8")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for name, (k, n) in SHAPES.items():
        for m in [int(x) for x in args.ms.split(",")]:
            for kk in ("NMV_MM_NT", "NMV_MM_SPLITS"):
                os.environ.pop(kk, None)
            os.environ["NMV_MM_WIDE"] = "0"
            us0, _ = bench(k, n, m, dev, args.dtype == "fp8")
            os.environ["NMV_MM_WIDE"] = "1"
            us1, _ = bench(k, n, m, dev, args.dtype == "fp8")
            res = [f"k-split {us0:.1f}", f"wide(plan) {us1:.1f}"]
            for form in args.forms.split(","):
                nt, sp = form.split(":")
                if (k // 256) % int(sp) or k // int(sp) < 512:
                    continue
                os.environ["NMV_MM_NT"], os.environ["NMV_MM_SPLITS"] = nt, sp
                us, _ = bench(k, n, m, dev, args.dtype == "fp8")
                res.append(f"nt{nt}/sp{sp} {us:.1f}")
            print(f"{args.dtype} {name:8s} M={m:3d} | " + "  ".join(res), flush=True)
