import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_magic_vllm_amd import _custom_ops as ops
dev = torch.device("cuda:0")
def timeit(fn, iters=40):
    fn(0); torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): fn(0)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters): fn(i)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (k, n) in ((4096, 4096), (14336, 4096)):
    ncopy = max(2, (600 << 20) // (k * n // 2))
    g = torch.Generator(device=dev).manual_seed(0)
    qs = [torch.randint(-2**31, 2**31 - 1, (k // 16, n * 2), dtype=torch.int32, device=dev, generator=g) for _ in range(ncopy)]
    sc = (torch.rand((k // 128, n), device=dev, generator=g) * 0.01).to(torch.bfloat16)
    w = torch.ones(n, dtype=torch.bfloat16, device=dev)
    e = torch.empty(0, dtype=torch.int32, device=dev)
    ws = torch.zeros(n // 64 * 16, dtype=torch.int32, device=dev)
    for m in (1, 8, 64):
        a = (torch.randn((m, k), device=dev, generator=g) * 0.1).to(torch.bfloat16)
        res = torch.zeros((m, n), dtype=torch.bfloat16, device=dev)
        def plain(i):
            out = ops.gptq_marlin_gemm(a, qs[i % ncopy], sc, e, e, ws, 4, m, n, k, True)
            ops.fused_add_rms_norm(out, res, w, 1e-5)
        def gemm_only(i):
            ops.gptq_marlin_gemm(a, qs[i % ncopy], sc, e, e, ws, 4, m, n, k, True)
        def part_only(i):
            ops.gptq_marlin_gemm_partial(a, qs[i % ncopy], sc, m, n, k)
        def deferred(i):
            slab = ops.gptq_marlin_gemm_partial(a, qs[i % ncopy], sc, m, n, k)
            ops.fused_add_rms_norm_partial(slab, res, w, 1e-5)
        print(f"k={k} n={n} M={m} splits={ops.gptq_marlin_gemm_partial_splits(m, n, k)}: gemm {timeit(gemm_only):.1f}  partial {timeit(part_only):.1f} | gemm+norm {timeit(plain):.1f}  partial+norm {timeit(deferred):.1f} us", flush=True)
