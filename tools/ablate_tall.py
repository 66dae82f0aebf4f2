"""GPU, development build only (make ... EXTRA=-DNMV_W4_ABLATION; NMV_HIP_LIB=build/libnmvllm_hip_abl.so):
where the tall kernel's time goes at M = 64 (two 32-row blocks vs one 64-row tile) and at M = 16."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_gemm  # noqa: E402

NAMES = {0: "full", 1: "-mfma", 2: "-A", 4: "-flush", 6: "-A-flush", 10: "-A-barrier", 14: "-A-flush-barrier",
         15: "W loads only"}
KN = ("NMV_W4_TALL_MT", "NMV_W4_TALL_WK", "NMV_W4_SPLITS", "NMV_W4_DBG")

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    for name in ("gate_up", "down", "qkv"):
        k, n = bench_gemm.SHAPES[name]
        for m, mts in ((64, (2, 4)), (16, (1, ))):
            for mt in mts:
                for wk in (2, 4):
                    row = []
                    for dbg in (0, 1, 2, 4, 6, 10, 14, 15):
                        for k_ in KN:
                            os.environ.pop(k_, None)
                        os.environ.update(NMV_W4_TALL_MT=str(mt), NMV_W4_TALL_WK=str(wk), NMV_W4_DBG=str(dbg))
                        try:
                            us, _ = bench_gemm.bench(name, k, n, m, dev, iters=24)
                            row.append(f"{NAMES[dbg]}={us:.1f}")
                        except Exception as e:
                            row.append(f"{NAMES[dbg]}=ERR")
                    print(f"{name:8s} M={m:2d} mt{mt} wk{wk}: " + "  ".join(row), flush=True)
