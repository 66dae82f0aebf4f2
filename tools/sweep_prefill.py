"""Times the prompt-sized W4A16 GEMM on the Llama-3-8B projections: the tall kernel on the Marlin tensor against the
prefill kernel on the native tensor (csrc/w4a16_prefill.hip) over its split-K counts; us per call from a hipGraph
replay, weights rotated through > 600 MB (tools/bench_gemm.py), and PFLOP/s.
usage: python tools/sweep_prefill.py [--ms 512,2048] [--shapes qkv,o,gate_up,down] [--splits 0,1,2,4]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_gemm import SHAPES, bench  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", default="512,2048")
    ap.add_argument("--shapes", default="qkv,o,gate_up,down")
    ap.add_argument("--splits", default="0,1,2,4,8", help="0 = the plan's own choice")
    ap.add_argument("--tiles", default="1,2", help="NMV_W4P_TILE: 0 = the plan's choice, 1 = 128 x 128, 2 = 256 x 256")
    ap.add_argument("--deferred", action="store_true", help="qkv / o / down as the model issues them: fp32 slabs only (mode 2)")
    ap.add_argument("--iters", type=int, default=8)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for name in args.shapes.split(","):
        k, n = SHAPES[name]
        for m in [int(x) for x in args.ms.split(",")]:
            md = 1 if name.startswith("gate_up") else (2 if args.deferred else 0)   # the step issues gate_up with the silu epilogue
            flop = 2.0 * m * k * n
            os.environ["NMV_W4P"] = "0"
            us, _ = bench(name, k, n, m, dev, iters=args.iters, native=None, mode=md)
            res = [f"tall(marlin) {us:.1f} ({flop / us / 1e9:.2f} PF/s)"]
            os.environ["NMV_W4P"] = "2"
            os.environ["NMV_W4P_MIN_M"] = "65"
            for tile in [int(x) for x in args.tiles.split(",")]:
                os.environ["NMV_W4P_TILE"] = str(tile)
                for sp in [int(x) for x in args.splits.split(",")]:
                    if sp and ((k // 128) % sp or (md == 1 and sp > 1)):
                        continue
                    os.environ["NMV_W4P_SPLITS"] = str(sp)
                    us, _ = bench(name, k, n, m, dev, iters=args.iters, native=md)
                    res.append(f"{('auto', 't128', 't256')[tile]}/sp{sp} {us:.1f} ({flop / us / 1e9:.2f})")
            print(f"{name:8s} M={m:4d} mode={md} | " + "  ".join(res), flush=True)
