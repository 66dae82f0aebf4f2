"""GPU: wall time of one prompt step (TTFT) beside the sum of its kernel times, so that launch overhead and kernel
time can be told apart.  Run it under `rocprofv3 --kernel-trace --stats` for the per-kernel table.
usage: python tools/profile_prefill.py [--prompt 512] [--batch 1] [--runs 8]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neural_magic_vllm_amd import _torch_bindings as tb  # noqa: E402
from neural_magic_vllm_amd.worker import decode_runner as dr  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--prompt", type=int, default=512)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--runs", type=int, default=8)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    quant = dict(method="gptq_marlin", bits=4, group_size=128)
    runner = dr.DecodeRunner(dr.LLAMA3_8B, dev, torch.bfloat16, quant, dr.CacheConfig(16, "auto"))
    runner.setup_batch(args.batch, args.prompt, 8)
    for _ in range(2):
        runner.prefill(args.prompt)
    torch.cuda.synchronize()
    wall, gpu = [], []
    for _ in range(args.runs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        runner.prefill(args.prompt)
        e1.record()
        t_issue = time.perf_counter()
        torch.cuda.synchronize()
        wall.append((time.perf_counter() - t0) * 1e3)
        gpu.append(e0.elapsed_time(e1))
        issue = (t_issue - t0) * 1e3
    wall.sort(), gpu.sort()
    print(f"binding={tb.binding} prompt={args.prompt} batch={args.batch}: wall p50 {wall[len(wall) // 2]:.2f} ms, "
          f"first-to-last kernel p50 {gpu[len(gpu) // 2]:.2f} ms, host issue time of the last run {issue:.2f} ms", flush=True)
