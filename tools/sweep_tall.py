"""GPU: plan sweep of the tall W4A16 kernel (and the MFMA-native one) at prefill-edge decode sizes: rows per
workgroup (mt), k groups (wk; columns per workgroup = 64 * 4 / wk) and split-K, in the self-contained form
(mode 0) and with the reduction deferred to the consumer (mode 2)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_gemm  # noqa: E402

KNOBS = ("NMV_W4_TALL_MT", "NMV_W4_TALL_WK", "NMV_W4_SPLITS")


def clear():
    for k in KNOBS:
        os.environ.pop(k, None)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", default="32,64")
    ap.add_argument("--shapes", default=bench_gemm.DEFAULT_SHAPES)
    ap.add_argument("--native", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    nat = (lambda md: md) if args.native else (lambda md: None)
    for name in args.shapes.split(","):
        k, n = bench_gemm.SHAPES[name]
        for m in [int(x) for x in args.ms.split(",")]:
            for md in (0, 2):
                clear()
                kw = dict(native=md) if args.native else dict(mode=md)
                base, _ = bench_gemm.bench(name, k, n, m, dev, iters=24, **kw)
                res = []
                for mt in (1, 2, 4, 8):
                    for wk in (4, 2, 1):
                        for sp in (1, 2, 3, 4, 8):
                            clear()
                            os.environ.update(NMV_W4_TALL_MT=str(mt), NMV_W4_TALL_WK=str(wk), NMV_W4_SPLITS=str(sp))
                            try:
                                us, _ = bench_gemm.bench(name, k, n, m, dev, iters=24, **kw)
                            except Exception:
                                continue
                            res.append((us, mt, wk, sp))
                clear()
                res.sort()
                best = ", ".join(f"mt{a}/wk{w}/sp{s}:{u:.1f}" for u, a, w, s in res[:8])
                print(f"{name:8s} M={m:3d} {'deferred' if md else 'plain   '} default {base:6.1f} us | best {best}",
                      flush=True)
