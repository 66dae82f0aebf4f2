"""bench.py -- decode tokens/s of Llama-3-8B w4a16 (GPTQ g128, marlin format) on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched
through torch.distributed.run with one rank per GPU (tensor parallel over RCCL/xGMI).  Rank 0
prints ONE JSON line.  A "step" is one decode step of the whole model for a batch of B
sequences (embedding, 32 decoder layers through the HIP kernels, bf16 lm_head, greedy sample),
replayed from a captured hipGraph; inputs (weights, KV cache with `context` tokens per
sequence) are resident in HBM before the timed region.

Extra objects on the line:
  roofline     -- dominant kernel = the W4A16 GEMM (w4a16_gemm_tall_kernel): algorithmic bytes of
                  the 4 GEMMs of a layer / mean device time of those 4 launches, measured live with
                  HIP events around a hipGraph replay of the same kernels (no host launch cost).
  ttft_ms_p50  -- p50 wall time of one 512-token prompt step (BASELINE metric's second half).
  cpu_baseline -- the oracle (oracle/oracle.c, OpenMP) on the host cores: one decoder layer's
                  dequant+GEMMs and paged attention for the same batch, scaled to a full step.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("NMV_BENCH_BATCH", 64)))
    ap.add_argument("--context", type=int, default=512)
    ap.add_argument("--model", default="llama3-8b")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--kv-cache-dtype", default="auto")
    ap.add_argument("--quant", default="w4a16", choices=["w4a16", "w8a8", "bf16"],
                    help="w4a16 = the BASELINE metric (configs[2]); w8a8 = configs[3]; bf16 = configs[1] "
                         "(unquantised: library GEMMs + our attention / cache / glue kernels)")
    return ap.parse_args()


@torch.inference_mode()
def gemm_roofline(runner, batch, dev, groups=32):
    """Device time of the dominant kernel: the 4 quantised GEMMs of one decoder layer at
    M = batch (each call = ONE kernel launch; the split-K reduction is inside it), `groups`
    launch groups rotating over all layers' weights (HBM, not the 256 MiB Infinity Cache), captured
    in one hipGraph and timed with HIP events on the replay stream -- no host launch cost in the
    number.  profiles/ holds the rocprofv3 kernel-trace of the same kernels (tools/gpu_ci.sh)."""
    from neural_magic_vllm_amd import _custom_ops as ops
    layers = runner.model.model.layers
    names = ("qkv_proj", "o_proj", "gate_up_proj", "down_proj")

    def mods_of(L):
        return [L.self_attn.qkv_proj, L.self_attn.o_proj, L.mlp.gate_up_proj, L.mlp.down_proj]

    mods = mods_of(layers[0])
    xs = [torch.randn((batch, m.input_size_per_partition), device=dev, dtype=runner.dtype)
          for m in mods]

    as_model = [False]  # second pass: the launches exactly as the model issues them

    def run(li, only=None):
        for j, (m, x) in enumerate(zip(mods_of(layers[li % len(layers)]), xs)):
            if only is not None and only != j:
                continue
            k, n = m.input_size_per_partition, m.output_size_per_partition
            if as_model[0] and getattr(m, "gate_up_interleaved", False):
                ops.gptq_marlin_gemm_silu_mul(x, m.qweight, m.scales, m.workspace, batch, n, k)
            elif as_model[0] and j != 2 and ops.gptq_marlin_gemm_partial_splits(batch, n, k) >= 1:
                ops.gptq_marlin_gemm_partial(x, m.qweight, m.scales, batch, n, k)
            else:
                # gate_up weights may be column-interleaved for the silu epilogue: same bytes, same time
                ops.gptq_marlin_gemm(x, m.qweight, m.scales, m.g_idx, m.g_idx_sort_indices,
                                     m.workspace, 4, batch, n, k, m.is_k_full)

    def timed(only):
        run(0, only)
        torch.cuda.synchronize(dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            run(0, only)
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for i in range(groups):
                run(i, only)
        graph.replay()
        torch.cuda.synchronize(dev)
        stream = torch.cuda.current_stream(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        graph.replay()
        e1.record(stream)
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / groups * 1e3  # us per group

    algs = []
    for m in mods:
        k, n = m.input_size_per_partition, m.output_size_per_partition
        algs.append(k * n // 2 + m.scales.numel() * 2 + 2 * batch * k + 2 * batch * n)
    alg = sum(algs)
    us = timed(None)
    per = {nm: {"us": round(timed(j), 2), "algorithmic_bytes": algs[j]} for j, nm in enumerate(names)}
    for v in per.values():
        v["GB/s"] = round(v["algorithmic_bytes"] / v["us"] / 1e3, 1)
    achieved = alg / (us * 1e-6) / 1e9
    in_model = None
    if os.environ.get("NMV_FUSED_GLUE", "1") != "0":
        as_model[0] = True
        in_model = {"avg_us_per_launch_group": round(timed(None), 2),
                    "per_gemm_us": {nm: round(timed(j), 2) for j, nm in enumerate(names)},
                    "note": "the same four launches as the decode step issues them: qkv / o / down leave "
                            "fp32 split-K slabs that the following rope+cache / norm launch sums (deferred "
                            "reduction), gate_up applies silu_and_mul in its epilogue; `achieved` above is "
                            "for the self-contained op (reduction inside the launch)"}
        as_model[0] = False
    # HBM traffic from the PMC pass (FETCH_SIZE/WRITE_SIZE, separate rocprofv3 runs, gfx950
    # correction applied): measured offline with tools/gpu_ci.sh, summary committed under profiles/
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "gemm_traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            t = json.load(f)
        traffic = t.get("by_batch", {}).get(str(batch))
    kern = "w4a16_gemm_tall_kernel"
    return {"bound": "hbm", "kernel": f"{kern} (the 4 GEMM launches of one decoder layer, M={batch})",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "algorithmic_bytes_per_launch_group": alg, "avg_us_per_launch_group": round(us, 2),
            "launches_per_group": 4, "timing": "HIP events around a hipGraph replay", "per_gemm": per,
            "in_model": in_model}


@torch.inference_mode()
def ttft(runner, dev, prompt_len, runs=5):
    """p50 time to first token: one prompt of `prompt_len` tokens through the whole model (our
    GEMM / norm / rope / cache-write / prompt flash-attention kernels, launched eagerly)."""
    runner.setup_batch(1, prompt_len, 8)
    ts = []
    for i in range(runs + 1):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        runner.prefill(prompt_len, seed=i)
        torch.cuda.synchronize(dev)
        ts.append((time.perf_counter() - t0) * 1e3)
    ts = sorted(ts[1:])
    return round(ts[len(ts) // 2], 3)


def cpu_baseline(arch, batch, context, budget_s=12.0):
    """the CPU oracle on the host cores: decoder-layer samples (4 dequant+GEMMs at M=batch and paged
    attention over `context` tokens), repeated until ~`budget_s` seconds of CPU work have been timed,
    extrapolated to the whole step (layers x; lm_head and glue not included)."""
    import helpers
    import oracle
    from oracle import ref_math
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    oracle.build()
    h, inter, hd = arch.hidden_size, arch.intermediate_size, arch.head_dim
    nq, nkv = arch.num_attention_heads, arch.num_key_value_heads
    shapes = [(h, (nq + 2 * nkv) * hd), (nq * hd, h), (h, 2 * inter), (inter, h)]
    g = torch.Generator().manual_seed(0)
    probs = []
    for k, n in shapes:
        q_w = torch.randint(0, 16, (k, n), generator=g, dtype=torch.int32)
        mq = ref_math.marlin_weights(q_w, k, n, 4)
        s = (torch.rand((k // 128, n), generator=g) * 0.01 + 0.001).to(torch.bfloat16)
        ms = ref_math.marlin_permute_scales(s, k, n, 128)
        a = torch.randn((batch, k), generator=g).to(torch.bfloat16)
        probs.append((a, mq, ms, n, k))
    nblk = batch * ((context + 15) // 16) + 8
    inp = helpers.make_paged_attention_inputs(0, batch, (nq, nkv), hd, 16, torch.bfloat16,
                                              seq_lens=[context] * batch, num_blocks=nblk)
    t_gemm = t_attn = 0.0
    layers = 0
    while layers == 0 or (t_gemm + t_attn < budget_s and layers < 64):
        for a, mq, ms, n, k in probs:
            t0 = time.perf_counter()
            oracle.gptq_marlin_gemm(a, mq, ms, None, None, 4, batch, n, k)
            t_gemm += time.perf_counter() - t0
        t0 = time.perf_counter()
        oracle.paged_attention(inp["query"], inp["key_cache"], inp["value_cache"], nkv, inp["scale"],
                               inp["block_tables"], inp["seq_lens"], 16)
        t_attn += time.perf_counter() - t0
        layers += 1
    step_s = (t_gemm + t_attn) / layers * arch.num_hidden_layers
    return {"value": round(batch / step_s, 3), "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": (f"oracle.c (OpenMP, {cores} threads) timed on {layers} decoder-layer samples: 4 "
                       f"dequant+GEMMs at M={batch} and paged attention over {context} tokens x {batch} "
                       f"seqs ({t_gemm:.2f}s + {t_attn:.2f}s of CPU work), scaled to "
                       f"{arch.num_hidden_layers} layers; lm_head and glue not included")}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # rehearsal knobs (tests/test_gpu_tp.py): all ranks on one GPU with gloo collectives, to exercise
    # this script's N > 1 path on a 1-GPU box; the measured configuration is one rank per GPU + RCCL
    backend = os.environ.get("NMV_BENCH_DIST_BACKEND", "nccl")
    if os.environ.get("NMV_BENCH_SINGLE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    import torch.distributed as dist
    from neural_magic_vllm_amd import distributed as nd
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
        nd.initialize_model_parallel(world, backend=backend, local_rank=local_rank)
    from neural_magic_vllm_amd.worker import decode_runner as dr
    arch = {"llama3-8b": dr.LLAMA3_8B, "llama3-70b": dr.LLAMA3_70B, "tiny": dr.TINY}[args.model]
    quant = {"w4a16": dict(method="gptq_marlin", bits=4, group_size=128),
             "w8a8": dict(method="w8a8", bits=8, group_size=-1), "bf16": None}[args.quant]
    runner = dr.DecodeRunner(arch, dev, torch.bfloat16, quant,
                             dr.CacheConfig(16, args.kv_cache_dtype))

    def measure(batch, steps, warmup):
        runner.setup_batch(batch, args.context, steps + warmup + 8)
        runner.fill_context()
        graphed = False if args.no_graph else runner.capture()
        if world > 1:  # every rank replays a graph, or none does
            ok = torch.tensor([1 if graphed else 0], device=dev, dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if graphed and int(ok.item()) == 0:
                runner.graph = None
                graphed = False
        for _ in range(warmup):
            runner.decode_step()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            runner.decode_step()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, graphed

    dt, graphed = measure(args.batch, args.steps, args.warmup)
    ms_per_step = dt / args.steps * 1e3
    value = args.batch * args.steps / dt

    out = {
        "metric": {"w4a16": "decode tokens/sec, Llama-3-8B w4a16 (GPTQ-marlin g128), bf16 activations",
                   "w8a8": "decode tokens/sec, Llama-3-8B w8a8 (int8 per-channel x dynamic per-token int8)",
                   "bf16": "decode tokens/sec, Llama-3-8B bf16 (unquantised)"}[args.quant],
        "value": round(value, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "int8" if args.quant == "w8a8" else "bf16",
        "data": {"w4a16": "synthetic (random-init N(0,0.02) weights quantised to int4 g128, random KV context)",
                 "w8a8": "synthetic (random-init N(0,0.02) weights quantised to int8 per channel, random KV context)",
                 "bf16": "synthetic (random-init N(0,0.02) bf16 weights, random KV context)"}[args.quant],
        "config": {"workload": f"{args.model} {args.quant} decode step, batch {args.batch}, "
                               f"context {args.context} tokens/seq, block 16, kv {args.kv_cache_dtype}",
                   "global_batch": args.batch, "context_len": args.context,
                   "parallelism": f"tp{world}", "hip_graph": graphed},
    }
    if world > 1:
        # which path carries the row-parallel all-reduces: the one-shot P2P kernel over HIP IPC (after its
        # start-up self-test on these devices) or the process group (RCCL)
        car = nd.get_tp_group().custom_ar
        out["config"]["all_reduce"] = "p2p one-shot over HIP IPC (fused with residual-add + RMSNorm)" \
            if car is not None else f"process group ({backend})"
    if rank == 0:
        wb = runner.weight_bytes_per_step()
        kv_elem = 1 if args.kv_cache_dtype.startswith("fp8") else 2
        kvb = 2 * args.context * runner.num_kv_heads * arch.head_dim * kv_elem * args.batch * arch.num_hidden_layers
        out["step_roofline"] = {"weight_bytes": wb, "kv_bytes": kvb,
                                "hbm_bound_ms": round((wb + kvb) / (HBM_PEAK_GBS * 1e9) * 1e3, 4),
                                "frac_of_hbm_bound": round((wb + kvb) / (HBM_PEAK_GBS * 1e9) / (ms_per_step * 1e-3), 4)}
        if args.quant == "w4a16":
            out["roofline"] = gemm_roofline(runner, args.batch, dev)
    if rank == 0 and world == 1 and not args.no_sweep:
        out["ttft_ms_p50"] = {"prompt_tokens": args.context, "batch": 1,
                              "value": ttft(runner, dev, args.context)}
    if world == 1 and not args.no_sweep:
        sweep = {}
        for b in (1, 8, 32):
            d, _ = measure(b, max(16, args.steps // 2), 4)
            sweep[str(b)] = round(b * max(16, args.steps // 2) / d, 1)
        sweep[str(args.batch)] = round(value, 1)
        out["batch_sweep_tokens_per_s"] = sweep
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(arch, args.batch, args.context)
        except Exception as e:  # the baseline must never take the GPU number down with it
            out["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": os.cpu_count(),
                                   "kind": "port", "sample": f"failed: {e!r}"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        nd.destroy_model_parallel()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
