"""host enqueue time against total time of one 512-token prompt step (is TTFT host-bound?)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neural_magic_vllm_amd.worker import decode_runner as dr  # noqa: E402
dev = torch.device("cuda:0")
runner = dr.DecodeRunner(dr.LLAMA3_8B, dev, torch.bfloat16, dict(method="gptq_marlin", bits=4, group_size=128), dr.CacheConfig(16, "auto"))
runner.setup_batch(1, 512, 8)
for i in range(8):
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    runner.prefill(512, seed=i)
    t1 = time.perf_counter()
    torch.cuda.synchronize(dev)
    t2 = time.perf_counter()
    print(f"run {i}: host enqueue {1e3 * (t1 - t0):.2f} ms, total {1e3 * (t2 - t0):.2f} ms", flush=True)
