"""Times nmv_w4_native_gemm on the Llama-3-8B projections with the ring kernel (csrc/w4a16_ring.hip) off / on and
over its split-K counts: us per call from a hipGraph replay, weights rotated through > 600 MB (tools/bench_gemm.py).
usage: python tools/sweep_ring.py [--ms 33,64] [--shapes qkv,o,gate_up,down] [--modes 0,2]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_gemm import SHAPES, bench  # noqa: E402
from neural_magic_vllm_amd import _lib  # noqa: E402

KNOBS = ("NMV_W4R", "NMV_W4R_MIN_M", "NMV_W4R_MIN_WGS", "NMV_W4R_SPLITS", "NMV_W4R_MAX_SPLITS", "NMV_W4R_MT")

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", default="33,64")
    ap.add_argument("--shapes", default="qkv,o,gate_up,down")
    ap.add_argument("--modes", default="0,2")
    ap.add_argument("--splits", default="1,2,4,7,8,14,16")
    ap.add_argument("--iters", type=int, default=16)
    ap.add_argument("--mt", default="", help="force the row tile (NMV_W4R_MT): 2 = row blocks of 32, 4 = 64")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    for name in args.shapes.split(","):
        k, n = SHAPES[name]
        for m in [int(x) for x in args.ms.split(",")]:
            for mode in [int(x) for x in args.modes.split(",")]:
                md = 1 if (mode == 0 and name.startswith("gate_up")) else mode   # the step issues gate_up with the silu epilogue
                res = []
                for kk in KNOBS:
                    os.environ.pop(kk, None)
                os.environ["NMV_W4R"] = "0"
                if m <= 64:
                    us, _ = bench(name, k, n, m, dev, iters=args.iters, native=md)
                    res.append(f"stream {us:.1f}")
                else:   # prompt-sized: the tall kernel on the Marlin tensor is what the ring replaces
                    us, _ = bench(name, k, n, m, dev, iters=args.iters, native=None, mode=md)
                    res.append(f"tall(marlin) {us:.1f}")
                os.environ["NMV_W4R"] = "1"
                os.environ["NMV_W4R_MIN_M"] = "17"
                if args.mt:
                    os.environ["NMV_W4R_MT"] = args.mt
                os.environ["NMV_W4R_MIN_WGS"] = "1"
                os.environ["NMV_W4R_PREFILL"] = "1"
                for sp in [int(x) for x in args.splits.split(",")]:
                    if (k // 128) % sp or (k // 128) // sp > 32 or (md == 1 and sp > 1):
                        continue
                    os.environ["NMV_W4R_SPLITS"] = str(sp)
                    os.environ["NMV_W4R_MAX_SPLITS"] = "64"
                    us, _ = bench(name, k, n, m, dev, iters=args.iters, native=md)
                    res.append(f"ring/sp{sp} {us:.1f}")
                print(f"{name:8s} M={m:3d} mode={md} | " + "  ".join(res) + f"  | timeouts {lib.nmv_w4_ring_timeouts()}", flush=True)
