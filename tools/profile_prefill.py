"""a few 512-token prompt steps of Llama-3-8B w4a16 (the TTFT half of the metric) for rocprofv3 --kernel-trace --stats:
tools/profile_prefill.sh copies the kernel-stats summary to profiles/.  usage: profile_prefill.py [prompt_len] [runs]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neural_magic_vllm_amd.worker import decode_runner as dr  # noqa: E402

if __name__ == "__main__":
    prompt_len = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    runs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    dev = torch.device("cuda:0")
    quant = dict(method="gptq_marlin", bits=4, group_size=128)
    runner = dr.DecodeRunner(dr.LLAMA3_8B, dev, torch.bfloat16, quant, dr.CacheConfig(16, "auto"))
    runner.setup_batch(1, prompt_len, 8)
    ts = []
    for i in range(runs):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        runner.prefill(prompt_len, seed=i)
        torch.cuda.synchronize(dev)
        ts.append(round((time.perf_counter() - t0) * 1e3, 3))
    print("prompt", prompt_len, "wall ms per run:", ts, flush=True)
