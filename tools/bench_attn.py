"""GPU microbenchmark of paged_attention_v1/v2 at Llama-3-8B geometry (32 q heads, 8 kv heads,
D=128, block 16), hipGraph-timed; KV caches sized past the Infinity Cache and block tables random.
usage: python tools/bench_attn.py [--cases B:L,B:L,...] [--kv auto|fp8]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neural_magic_vllm_amd import _custom_ops as ops  # noqa: E402
from neural_magic_vllm_amd.attention.ops.paged_attn import PagedAttention  # noqa: E402


def bench(b, L, dev, kv, iters=20, nq=32, nkv=8, d=128, bs=16, fused=0, sparse=None, v2=False):
    """fused = S > 0: the decode step's form -- the kernel starts from S fp32 split-K slabs of the qkv projection
    (slab sums + rotary + cache store of the new token in its prologue; L counts the new token);
    sparse = (local_blocks, vert_stride, block_size, head_sliding_step): block-sparse attention"""
    blocks_per_seq = (L + bs - 1) // bs
    nb = max(b * blocks_per_seq * 2, (640 << 20) // (2 * nkv * d * bs * (1 if kv == "fp8" else 2)))
    cdt = torch.uint8 if kv == "fp8" else torch.bfloat16
    x = 16 // (1 if kv == "fp8" else 2)
    g = torch.Generator(device=dev).manual_seed(0)
    if kv == "fp8":
        kc = torch.randint(0, 120, (nb, nkv, d // x, bs, x), dtype=torch.uint8, device=dev, generator=g)
        vc = torch.randint(0, 120, (nb, nkv, d, bs), dtype=torch.uint8, device=dev, generator=g)
    else:
        kc = (torch.rand((nb, nkv, d // x, bs, x), device=dev, generator=g) - 0.5).to(cdt)
        vc = (torch.rand((nb, nkv, d, bs), device=dev, generator=g) - 0.5).to(cdt)
    q = torch.randn((b, nq, d), device=dev, dtype=torch.bfloat16, generator=g) * 0.1
    out = torch.empty_like(q)
    seq_lens = torch.full((b, ), L, dtype=torch.int32, device=dev)
    # distinct random blocks per (iteration, seq): every replayed call reads fresh HBM lines
    tables = [torch.randperm(nb, device=dev, generator=g)[:b * blocks_per_seq].to(torch.int32)
              .view(b, blocks_per_seq) for _ in range(iters)]

    if fused:
        slab = torch.randn((fused, b, (nq + 2 * nkv) * d), device=dev, dtype=torch.float32, generator=g) * 0.05
        positions = torch.full((b, ), L - 1, dtype=torch.int64, device=dev)
        t = torch.arange(8192, device=dev, dtype=torch.float32)[:, None] * \
            (10000.0 ** (-torch.arange(0, d, 2, device=dev, dtype=torch.float32) / d))[None, :]
        cos_sin = torch.cat([t.cos(), t.sin()], dim=-1).to(torch.bfloat16).contiguous()
        slots = [(tb[:, (L - 1) // bs].to(torch.int64) * bs + (L - 1) % bs).contiguous() for tb in tables]

    bufs = None
    if v2:   # the partitioned form (512-token partitions + the reduce launch)
        from neural_magic_vllm_amd.attention.ops.paged_attn import _PartitionBuffers
        bufs = _PartitionBuffers(b, nq, d, L, torch.bfloat16, dev).as_tuple()

    def run(i):
        if fused:
            ops.paged_attention_rope_partial(out, slab, positions, cos_sin, slots[i], kc, vc, nq, nkv, d, d**-0.5,
                                             tables[i], seq_lens, bs, L, "fp8" if kv == "fp8" else "auto", 1.0, bufs)
        elif v2:
            ops.paged_attention_v2(out, *bufs, q, kc, vc, nkv, d**-0.5, tables[i], seq_lens, bs, L, None,
                                   "fp8" if kv == "fp8" else "auto", 1.0)
        elif sparse:
            ops.paged_attention_v1(out, q, kc, vc, nkv, d**-0.5, tables[i], seq_lens, bs, L, None,
                                   "fp8" if kv == "fp8" else "auto", 1.0, 0, sparse[0], sparse[1], sparse[2], sparse[3])
        else:
            ops.paged_attention_v1(out, q, kc, vc, nkv, d**-0.5, tables[i], seq_lens, bs, L, None,
                                   "fp8" if kv == "fp8" else "auto", 1.0)

    run(0)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run(0)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for i in range(iters):
            run(i)
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    graph.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    alg = 2 * b * L * nkv * d * (1 if kv == "fp8" else 2)
    return us, alg / us / 1e3


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="64:512,32:512,8:512,1:512,64:2048,8:8192")
    ap.add_argument("--kv", default="auto")
    ap.add_argument("--fused", type=int, default=0, help="also time the fused rope + cache + attention form from "
                    "this many fp32 qkv slabs")
    ap.add_argument("--v2", action="store_true", help="also time the partitioned form (v2 + reduce) of every column")
    ap.add_argument("--sparse", default="", help="local_blocks,vert_stride,block_size,head_sliding_step: also time "
                    "block-sparse attention (masked windows are skipped before their K / V are loaded)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for c in args.cases.split(","):
        b, L = (int(v) for v in c.split(":"))
        us, gbs = bench(b, L, dev, args.kv)
        extra = ""
        if args.sparse:
            su, _ = bench(b, L, dev, args.kv, sparse=tuple(int(v) for v in args.sparse.split(",")))
            extra += f"   block-sparse ({args.sparse}) {su:8.1f} us"
        if args.v2:
            vu, _ = bench(b, L, dev, args.kv, v2=True)
            extra += f"   v2 {vu:8.1f} us"
        if args.fused:
            fu, fg = bench(b, L, dev, args.kv, fused=args.fused)
            extra += f"   fused prologue ({args.fused} slabs) {fu:8.1f} us {fg:7.0f} GB/s"
            if args.v2:
                fv, _ = bench(b, L, dev, args.kv, fused=args.fused, v2=True)
                extra += f"   fused v2 {fv:8.1f} us"
        print(f"attn v1 B={b:3d} L={L:5d} kv={args.kv}  {us:8.1f} us  {gbs:7.0f} GB/s{extra}", flush=True)


def bench_prefill(L, nseq, dev, iters=10, nq=32, nkv=8, d=128):
    """prompt attention: `nseq` prompts of L tokens; compares with torch SDPA on the same data"""
    t = L * nseq
    qkv = torch.randn((t, (nq + 2 * nkv) * d), device=dev, dtype=torch.bfloat16)
    q, k, v = qkv.split([nq * d, nkv * d, nkv * d], dim=-1)
    q, k, v = q.view(t, nq, d), k.view(t, nkv, d), v.view(t, nkv, d)
    out = torch.empty((t, nq, d), device=dev, dtype=torch.bfloat16)
    cu = torch.arange(0, t + 1, L, dtype=torch.int32, device=dev)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    us = timed(lambda: ops.prefill_attention(out, q, k, v, cu, L, d**-0.5))

    def sdpa():
        for s in range(nseq):
            qs = q[s * L:(s + 1) * L].movedim(0, 1)
            ks = k[s * L:(s + 1) * L].movedim(0, 1).repeat_interleave(nq // nkv, dim=0)
            vs = v[s * L:(s + 1) * L].movedim(0, 1).repeat_interleave(nq // nkv, dim=0)
            torch.nn.functional.scaled_dot_product_attention(qs, ks, vs, is_causal=True, scale=d**-0.5)

    us_sdpa = timed(sdpa)
    flops = 4.0 * nseq * L * L * nq * d / 2
    return us, flops / us / 1e6, us_sdpa


if __name__ == "__main__" and os.environ.get("NMV_BENCH_PREFILL"):
    dev = torch.device("cuda:0")
    for L, nseq in [(512, 1), (512, 8), (2048, 1), (8192, 1)]:
        us, tf, us_sdpa = bench_prefill(L, nseq, dev)
        print(f"prefill attn L={L:5d} x{nseq}  {us:9.1f} us  {tf:6.1f} TFLOP/s   torch SDPA {us_sdpa:9.1f} us", flush=True)
