"""GPU, development build only (make ... EXTRA=-DNMV_W4_ABLATION; NMV_HIP_LIB=build/libnmvllm_hip_abl.so):
where the direct decode kernel's time goes -- the same launch with parts of the wave program removed."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_gemm  # noqa: E402

NAMES = {0: "full", 1: "no expand/MFMA", 2: "no A loads", 3: "loads of W only", 4: "no flush", 6: "no A, no flush"}

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    for name in ("gate_up", "down", "qkv"):
        k, n = bench_gemm.SHAPES[name]
        for m in (1, 16):
            for k_ in ("NMV_W4_DIRECT", "NMV_W4_DIRECT_WK", "NMV_W4_SPLITS", "NMV_W4_DBG"):
                os.environ.pop(k_, None)
            tall, _ = bench_gemm.bench(name, k, n, m, dev, iters=24)
            line = [f"{name:8s} M={m:2d} tall {tall:5.1f}"]
            for wk in (4, 8):
                for sp in ((1, ) if name == "gate_up" else (1, 2, 4)):
                    row = []
                    for dbg in (0, 1, 2, 3, 4, 6):
                        os.environ.update(NMV_W4_DIRECT="1", NMV_W4_DIRECT_WK=str(wk), NMV_W4_SPLITS=str(sp),
                                          NMV_W4_DBG=str(dbg))
                        try:
                            us, _ = bench_gemm.bench(name, k, n, m, dev, iters=24)
                            row.append(f"{NAMES[dbg]}={us:.1f}")
                        except Exception as e:
                            row.append(f"{NAMES[dbg]}=ERR")
                    line.append(f"\n      wk{wk}/sp{sp}: " + "  ".join(row))
            print("".join(line), flush=True)
