"""Sweep the plan knobs of w4a16_stream_kernel (NMV_W4S_* environment overrides, read per call) on the Llama-3-8B
projection shapes: us per call from a hipGraph replay with weights rotated through > 600 MB (tools/bench_gemm.py).
usage: python tools/sweep_stream.py [--ms 1,16,64] [--shapes qkv,o,gate_up,down] [--mode 0|1|2]"""
import argparse
import itertools
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_gemm import SHAPES, bench  # noqa: E402

KNOBS = ("NMV_W4S", "NMV_W4S_STRICT", "NMV_W4S_MT", "NMV_W4S_NW", "NMV_W4S_CPW", "NMV_W4S_D", "NMV_W4S_SPLITS", "NMV_W4S_GST")


def clear():
    for k in KNOBS:
        os.environ.pop(k, None)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", default="1,16,64")
    ap.add_argument("--shapes", default="qkv,o,gate_up,down")
    ap.add_argument("--mode", type=int, default=0)
    ap.add_argument("--iters", type=int, default=16)
    ap.add_argument("--native", action="store_true", help="nmv_w4_native_gemm (the MFMA-native tensor) in the same mode")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for name in args.shapes.split(","):
        k, n = SHAPES[name]
        for m in [int(x) for x in args.ms.split(",")]:
            res = []
            clear()
            os.environ["NMV_W4S"] = "0"
            us, _ = bench(name, k, n, m, dev, iters=args.iters, mode=args.mode, native=args.mode if args.native else None)
            res.append((us, "tall"))
            clear()
            us, _ = bench(name, k, n, m, dev, iters=args.iters, mode=args.mode, native=args.mode if args.native else None)
            res.append((us, "default"))
            mts = [1] if m <= 16 else ([2, 1] if m <= 32 else [4, 2, 1])
            for mt in mts:
                gsts = [0] if mt == 1 else ([0, 2] if mt == 2 else [1])
                for gst, nw, cpw, d, sp in itertools.product(gsts, (8, 16, 4), (1, 2, 4), (1, 2, 3), (1, 2, 4, 7, 8, 14)):
                    if gst and (nw != 8 or d > 2):
                        continue
                    if nw == 16 and (mt != 1 or cpw == 1 and d > 3):
                        continue
                    if nw == 4 and (mt != 1 or cpw == 4 or d != 3):
                        continue
                    clear()
                    os.environ.update({"NMV_W4S_STRICT": "1", "NMV_W4S_MT": str(mt), "NMV_W4S_NW": str(nw), "NMV_W4S_CPW": str(cpw),
                                       "NMV_W4S_D": str(d), "NMV_W4S_SPLITS": str(sp), "NMV_W4S_GST": str(gst)})
                    try:
                        us, _ = bench(name, k, n, m, dev, iters=args.iters, mode=args.mode, native=args.mode if args.native else None)
                    except Exception as e:  # plan not available / no kernel for it
                        continue
                    res.append((us, f"mt{mt}/gst{gst}/nw{nw}/cpw{cpw}/d{d}/sp{sp}"))
            clear()
            res.sort()
            print(f"{name:8s} M={m:3d} mode={args.mode}  tall {dict((b, a) for a, b in res)['tall']:.1f}  default "
                  f"{dict((b, a) for a, b in res)['default']:.1f} | " + "  ".join(f"{lab}:{u:.1f}" for u, lab in res[:10]), flush=True)
