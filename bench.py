"""bench.py -- decode tokens/s of Llama-3-8B w4a16 (GPTQ g128, marlin format) on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched
through torch.distributed.run with one rank per GPU (tensor parallel over RCCL/xGMI) -- and when it
is started directly (`python bench.py --gpus N`, no WORLD_SIZE in the environment) it starts those N
ranks itself, as a child `python -m torch.distributed.run`, before anything touches the GPU.  Every
rank checks WORLD_SIZE == --gpus and exits non-zero otherwise.  Rank 0 prints ONE JSON line.  A "step" is one decode step of the whole model for a batch of B
sequences (embedding, 32 decoder layers through the HIP kernels, bf16 lm_head, greedy sample),
replayed from a captured hipGraph; inputs (weights, KV cache with `context` tokens per
sequence) are resident in HBM before the timed region.

Extra objects on the line:
  roofline     -- dominant kernel = the W4A16 GEMM (w4a16_stream_kernel at M <= 64): algorithmic bytes of
                  the 4 GEMMs of a layer / mean device time of those 4 launches, measured live with
                  HIP events around a hipGraph replay of the same kernels (no host launch cost).
  ttft_ms_p50  -- p50 wall time of one 512-token prompt step (BASELINE metric's second half).
  cpu_baseline -- what the reference's CPU executor would run for the same step, timed on this host's
                  cores: weights dequantised once to bf16 + F.linear (oneDNN) for the 4 GEMMs of a layer,
                  and the reference's own CPU paged-attention kernel (oracle/_ref) or its restatement
                  (oracle.c), on a bounded sample scaled to the whole step.
A P2P all-reduce whose flag wait timed out (a lost peer) makes the run exit non-zero without a JSON line.
"""
import subprocess
import socket
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
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("NMV_BENCH_BATCH", 64)))
    ap.add_argument("--context", type=int, default=512)
    ap.add_argument("--model", default="llama3-8b")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--e2e", action="store_true",
                    help="add `e2e`: prompt 512 -> 128 greedy tokens for batch 1..64, 3 + 10 runs (SURVEY 8d); ~1 min more")
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
    marlin_cache = {}

    def marlin_like(li, j, k, n):
        key = (li, j)
        if key not in marlin_cache:
            g = torch.Generator(device=dev).manual_seed(1000 + 7 * li + j)
            marlin_cache[key] = (torch.randint(-2**31, 2**31 - 1, (k // 16, n * 2), dtype=torch.int32, device=dev, generator=g),
                                 (torch.rand((k // 128, n), device=dev, generator=g) * 0.01).to(runner.dtype))
        return marlin_cache[key]

    def run(li, only=None):
        for j, (m, x) in enumerate(zip(mods_of(layers[li % len(layers)]), xs)):
            if only is not None and only != j:
                continue
            k, n = m.input_size_per_partition, m.output_size_per_partition
            if as_model[0] and getattr(m, "gate_up_interleaved", False):
                m.quant_method.apply_silu_mul(m, x)     # the LinearMethod's own entry points: MFMA-native copy when it keeps one
            elif as_model[0] and j != 2 and m.quant_method.can_defer(m, batch):
                m.quant_method.apply_partial(m, x)
            else:
                # the reference op on a Marlin tensor of the same shape (random codes: same bytes, same time -- the layers
                # themselves keep only the native tensor)
                qw, sc = marlin_like(li % len(layers), j, k, n)
                ops.gptq_marlin_gemm(x, qw, sc, m.g_idx, m.g_idx_sort_indices, m.workspace, 4, batch, n, k, m.is_k_full)

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

    for li in range(min(groups, len(layers))):      # built before any capture: a generator cannot be made inside one
        for j, m in enumerate(mods_of(layers[li])):
            marlin_like(li, j, m.input_size_per_partition, m.output_size_per_partition)
    algs = []
    for m in mods:
        k, n = m.input_size_per_partition, m.output_size_per_partition
        algs.append(k * n // 2 + (k // 128) * n * 2 + 2 * batch * k + 2 * batch * n)
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
                    "weights": "MFMA-native copy (nmv_w4_native_gemm)" if getattr(mods[0], "qweight_native", None) is not None
                               else "Marlin tensor",
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
    # The reference op (gptq_marlin_gemm on the Marlin interchange tensor, split-K reduced inside the launch) measured
    # above; the kernel the decode step actually spends its time in is the same GEMM in the forms the step issues
    # (MFMA-native tensor; qkv / o / down leave fp32 slabs to the next launch, gate_up applies silu * up), so THAT launch
    # group is the `roofline` of this line and the reference op sits beside it.
    ref_op = {"kernel": "w4a16_stream_kernel on the Marlin tensor (gptq_marlin_gemm, reduction inside the launch)",
              "achieved": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBS, 4),
              "avg_us_per_launch_group": round(us, 2), "per_gemm": per,
              "traffic": traffic["bytes"] if traffic else None, "traffic_detail": traffic,
              "traffic_source": "profiles/gemm_traffic.json (builder-run rocprofv3 PMC passes of round 3, not measured in this run)"}
    if in_model is None:
        return {"bound": "hbm", "kernel": ref_op["kernel"] + f", 4 launches of one decoder layer, M={batch}",
                "achieved": ref_op["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ref_op["frac"],
                "traffic": ref_op["traffic"], "traffic_unit": "bytes per launch group",
                "traffic_source": ref_op["traffic_source"], "algorithmic_bytes_per_launch_group": alg,
                "avg_us_per_launch_group": round(us, 2), "launches_per_group": 4,
                "timing": "HIP events around a hipGraph replay", "per_gemm": per}
    # algorithmic bytes of the step's forms: the fused gate_up writes [M, N / 2]
    alg_step = alg - batch * mods[2].output_size_per_partition
    step_us = in_model["avg_us_per_launch_group"]
    step_ach = alg_step / (step_us * 1e-6) / 1e9
    st = None
    spath = os.path.join(ROOT, "profiles", "r04_step_gemm_traffic.json")
    if os.path.exists(spath):
        with open(spath) as f:
            st = json.load(f).get(str(batch), {}).get("_group")
    ring = 17 <= batch <= 64 and os.environ.get("NMV_W4R", "1") != "0"
    kern = ("w4a16_ring_kernel (gate_up" + (", qkv, o, down" if batch <= 32 else "") + ")"
            + (" + w4a16_stream_kernel (qkv, o, down)" if batch > 32 else "")) if ring else "w4a16_stream_kernel"
    return {"bound": "hbm",
            "kernel": f"{kern} on the MFMA-native tensor: the 4 GEMM launches of one decoder layer as the decode step "
                      f"issues them (qkv / o / down deferred split-K, gate_up with the silu * up epilogue), M={batch}",
            "achieved": round(step_ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(step_ach / HBM_PEAK_GBS, 4),
            "traffic": st["bytes"] if st else None, "traffic_unit": "bytes per launch group",
            "traffic_source": "profiles/r04_step_gemm_traffic.json (builder-run rocprofv3 FETCH_SIZE / WRITE_SIZE passes of "
                              "tools/bench_step_gemms.py, not measured in this run)" if st else None,
            "traffic_ratio_to_algorithmic": st["ratio_to_algorithmic"] if st else None,
            "algorithmic_bytes_per_launch_group": alg_step, "avg_us_per_launch_group": step_us,
            "launches_per_group": 4, "timing": "HIP events around a hipGraph replay",
            "per_gemm_us": in_model["per_gemm_us"], "weights": in_model["weights"],
            "reference_op": ref_op}


MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA


@torch.inference_mode()
def prefill_gemm_roofline(runner, prompt_len, dev, groups=8):
    """the prompt step's dominant kernels: the 4 W4A16 GEMMs of a decoder layer at M = prompt_len exactly as the model
    issues them there (qkv / o / down leave fp32 slabs for the following launch when the layer defers, gate_up carries
    silu * up), `groups` launch groups rotating over the layers' weights, one hipGraph, HIP events: MFMA-bound,
    2 M N K flops per GEMM against the dense bf16 peak"""
    layers = runner.model.model.layers

    def mods_of(L):
        return [L.self_attn.qkv_proj, L.self_attn.o_proj, L.mlp.gate_up_proj, L.mlp.down_proj]

    mods = mods_of(layers[0])
    xs = [torch.randn((prompt_len, m.input_size_per_partition), device=dev, dtype=runner.dtype) for m in mods]

    def run(li, only=None):
        for j, (m, x) in enumerate(zip(mods_of(layers[li % len(layers)]), xs)):
            if only is not None and only != j:
                continue
            if getattr(m, "gate_up_interleaved", False):
                m.quant_method.apply_silu_mul(m, x)
            elif j != 2 and m.quant_method.can_defer(m, prompt_len):
                m.quant_method.apply_partial(m, x, True)    # as LlamaAttention / LlamaMLP call it at tp = 1: slabs in the model dtype
            else:
                m.quant_method.apply(m, x)

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

    flops = [2.0 * prompt_len * m.input_size_per_partition * m.output_size_per_partition for m in mods]
    us = timed(None)
    per = {nm: round(timed(j), 2) for j, nm in enumerate(("qkv_proj", "o_proj", "gate_up_proj", "down_proj"))}
    achieved = sum(flops) / (us * 1e-6) / 1e12
    return {"bound": "mfma", "kernel": "w4a16_prefill_kernel / w4a16_prefill_small_kernel on the MFMA-native tensor: the 4 GEMM launches "
                                        f"of one decoder layer at M = {prompt_len} as the prompt step issues them",
            "achieved": round(achieved, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
            "flops_per_launch_group": int(sum(flops)), "avg_us_per_launch_group": round(us, 2), "per_gemm_us": per,
            "timing": "HIP events around a hipGraph replay"}


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


@torch.inference_mode()
def e2e_latency(runner, dev, prompt_len=512, output_len=128, batches=(1, 2, 4, 8, 16, 32, 64), warm=3, timed=10):
    """SURVEY.md section 8(d) / the reference's benchmarks/benchmark_latency.py: `batch` prompts of `prompt_len` tokens,
    `output_len` greedy tokens each; 3 warm-up + 10 timed runs per batch size.  TTFT = wall time of the prompt step
    (p50 over the runs); decode tokens/s = batch * (output_len - 1) / (wall time of the output_len - 1 decode steps,
    replayed from a hipGraph captured once per batch size)."""
    out = {}
    for b in batches:
        runner.setup_batch(b, prompt_len, output_len + 8)
        graphed = runner.capture()
        state = [t.clone() for t in (runner.input_ids, runner.positions, runner.seq_lens, runner.slot_mapping)]
        steps_left = runner._steps_left
        ttfts, decs = [], []
        for i in range(warm + timed):
            for dst, src in zip((runner.input_ids, runner.positions, runner.seq_lens, runner.slot_mapping), state):
                dst.copy_(src)
            runner._steps_left = steps_left
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            first = runner.prefill(prompt_len, seed=i)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            runner.input_ids.copy_(first.view(-1))
            for _ in range(output_len - 1):
                runner.decode_step()
            torch.cuda.synchronize(dev)
            t2 = time.perf_counter()
            if i >= warm:
                ttfts.append((t1 - t0) * 1e3)
                decs.append(t2 - t1)
        ttfts.sort()
        out[str(b)] = {"ttft_ms_p50": round(ttfts[len(ttfts) // 2], 3),
                       "decode_tokens_per_s": round(b * (output_len - 1) * len(decs) / sum(decs), 1),
                       "decode_ms_per_step": round(sum(decs) / len(decs) / (output_len - 1) * 1e3, 4),
                       "hip_graph": graphed, "runs": len(decs)}
    return {"prompt_len": prompt_len, "output_len": output_len, "warmup_runs": warm, "timed_runs": timed, "by_batch": out}


def usable_cores() -> int:
    """CPU cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a GPU
    box hands a container a share of a 256-thread host; os.cpu_count() still says 256, and an OpenMP pool of
    that size on a 16-core quota runs ~20x slower than a pool of 16)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(arch, batch, context, budget_s=12.0):
    """The CPU path the reference would run for this step, on this host's cores: its CPU
    executor has no quantised GEMM, so a quantised model means weights dequantised ONCE to bf16 and
    F.linear (vllm/model_executor/layers/linear.py:103-136; oneDNN bf16) per projection, and its CPU
    paged-attention kernel (csrc/cpu/attention.cpp, compiled from the reference's sources into oracle/_ref
    when that build travelled here; else the restatement in oracle.c).  One decoder layer (4 GEMMs at
    M = batch + attention over batch x context tokens) is sampled until ~budget_s seconds of CPU work
    have been timed and scaled to the layer count; lm_head and glue are not included.  oracle.c's own
    (clarity-first) dequant-GEMM is timed once and reported separately."""
    import helpers
    import oracle
    from oracle import ref_math
    import torch.nn.functional as F
    cores = usable_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)
    torch.set_num_threads(cores)
    oracle.build()
    try:  # the OpenMP pools of oracle.c / oracle/_ref (libgomp reads OMP_NUM_THREADS once, at load)
        import ctypes
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(cores)
    except OSError:
        pass
    h, inter, hd = arch.hidden_size, arch.intermediate_size, arch.head_dim
    nq, nkv = arch.num_attention_heads, arch.num_key_value_heads
    shapes = [(h, (nq + 2 * nkv) * hd), (nq * hd, h), (h, 2 * inter), (inter, h)]
    g = torch.Generator().manual_seed(0)
    dense, t_deq = [], 0.0
    port_prob = None
    for k, n in shapes:
        q_w = torch.randint(0, 16, (k, n), generator=g, dtype=torch.int32)
        s = (torch.rand((k // 128, n), generator=g) * 0.01 + 0.001).to(torch.bfloat16)
        a = torch.randn((batch, k), generator=g).to(torch.bfloat16)
        t0 = time.perf_counter()
        # (q - 8) * s per group of 128, rounded to bf16: quantize_weights' w_ref (quant_utils.py:84-92)
        w = ((q_w - 8).to(torch.bfloat16).view(k // 128, 128, n) * s.view(k // 128, 1, n)).view(k, n)
        w_nk = w.t().contiguous()          # nn.Linear layout [out, in]
        t_deq += time.perf_counter() - t0
        dense.append((a, w_nk))
        if port_prob is None:
            port_prob = (a, ref_math.marlin_weights(q_w, k, n, 4), ref_math.marlin_permute_scales(s, k, n, 128), n, k)
    nblk = batch * ((context + 15) // 16) + 8
    inp = helpers.make_paged_attention_inputs(0, batch, (nq, nkv), hd, 16, torch.bfloat16,
                                              seq_lens=[context] * batch, num_blocks=nblk)
    attn_kind, attn = "port", None
    try:
        from oracle import build_ref
        if build_ref.load_ref():
            out = torch.empty_like(inp["query"])

            def attn():
                torch.ops._C_ref.paged_attention_v1(out, inp["query"], inp["key_cache"], inp["value_cache"], nkv,
                                                    inp["scale"], inp["block_tables"], inp["seq_lens"], 16,
                                                    inp["max_seq_len"], None, "auto", 1.0, 0, 0, 0, 64, 0)
            attn()
            attn_kind = "reference"
    except Exception:
        attn = None
    if attn is None:
        def attn():
            oracle.paged_attention(inp["query"], inp["key_cache"], inp["value_cache"], nkv, inp["scale"],
                                   inp["block_tables"], inp["seq_lens"], 16)
    for a, w_nk in dense:   # warm-up: oneDNN primitive creation, thread pool
        F.linear(a, w_nk)
    t_gemm = t_attn = 0.0
    layers = 0
    while layers == 0 or (t_gemm + t_attn < budget_s and layers < 2048):
        t0 = time.perf_counter()
        for a, w_nk in dense:
            F.linear(a, w_nk)
        t_gemm += time.perf_counter() - t0
        t0 = time.perf_counter()
        attn()
        t_attn += time.perf_counter() - t0
        layers += 1
    step_s = (t_gemm + t_attn) / layers * arch.num_hidden_layers
    # the restatement's own GEMM, for the record (one qkv projection)
    a, mq, ms, n, k = port_prob
    t0 = time.perf_counter()
    oracle.gptq_marlin_gemm(a, mq, ms, None, None, 4, batch, n, k)
    t_port = time.perf_counter() - t0
    return {"value": round(batch / step_s, 3), "unit": "tokens/s", "cores": cores, "kind": attn_kind,
            "sample": (f"{layers} decoder-layer samples on {cores} threads: F.linear bf16 (oneDNN) on weights "
                       f"dequantised once (4 GEMMs at M={batch}: {t_gemm / layers * 1e3:.2f} ms/layer; the one-time "
                       f"dequantisation of a layer took {t_deq:.2f} s, not counted) + "
                       f"{'the reference csrc/cpu paged_attention_v1 (oracle/_ref)' if attn_kind == 'reference' else 'oracle.c paged attention'}"
                       f" over {context} tokens x {batch} seqs ({t_attn / layers * 1e3:.2f} ms/layer); "
                       f"{t_gemm + t_attn:.1f} s of CPU work, scaled to {arch.num_hidden_layers} layers; lm_head and "
                       f"glue not included"),
            "oracle_c_gemm": {"qkv_proj_ms": round(t_port * 1e3, 1),
                              "note": "oracle.c's clarity-first dequant+GEMM (the parity checker), one call"}}


EXIT_CAPTURE_FAILED = 75   # EX_TEMPFAIL: the ranks ask to be restarted with --no-graph


def _capture_marker(env) -> str:
    """file a rank leaves behind when it exits because of a failed capture (torch.distributed.run does not pass the
    workers' exit code through); the parent of launch_ranks() names it through the environment"""
    return env.get("NMV_BENCH_CAPTURE_MARKER", "")


def weights_fit(arch, quant: str, world: int, batch: int, max_context: int, kv_dtype: str) -> dict:
    """bytes one rank holds for `arch` under tensor parallelism `world`: quantised decoder weights (int4 + group-128
    scales, int8 + channel scales, or bf16), bf16 embedding / lm_head shards, and the KV cache of `batch` sequences
    of `max_context` tokens -- against the 288 GB of an MI355X (BASELINE.json configs[4]: Llama-3-70B w4a16 at TP=8
    is 4.4 GB of weights per rank)"""
    h, inter, layers = arch.hidden_size, arch.intermediate_size, arch.num_hidden_layers
    kvh = max(arch.num_key_value_heads // world, 1)
    qkv = h * (arch.num_attention_heads // world + 2 * kvh) * arch.head_dim
    per_layer = qkv + (arch.num_attention_heads // world) * arch.head_dim * h + 3 * h * inter // world
    # w4a16: ONE tensor of codes + group-128 scales (GPTQMarlinLinearMethod keeps the MFMA-native tensor only); two when
    # NMV_W4_KEEP_MARLIN=1 keeps the Marlin tensor beside it
    both = os.environ.get("NMV_W4_NATIVE", "1") != "0" and os.environ.get("NMV_W4_KEEP_MARLIN", "0") == "1"
    bytes_per_w = {"w4a16": (0.5 + 2 / 128) * (2 if both else 1), "w8a8": 1.0, "bf16": 2.0}[quant]
    weights = layers * per_layer * bytes_per_w + 2 * (arch.vocab_size // world) * h * 2 + (2 * layers + 1) * h * 2
    kv = 2 * layers * batch * max_context * kvh * arch.head_dim * (1 if kv_dtype.startswith("fp8") else 2)
    total = weights + kv
    return {"weights_gb": weights / 1e9, "kv_gb": kv / 1e9, "total_gb": total / 1e9, "fits": total < 0.9 * 288e9}


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` started by hand: run the N ranks as a child torch.distributed.run (the
    command the driver itself uses) -- nothing in this process has touched the GPU yet, and it never
    will.  The ranks inherit stdout, so rank 0's JSON line is this command's JSON line."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["NMV_BENCH_CAPTURE_MARKER"] = os.path.join("/tmp", f"nmv_bench_capture_failed_{os.getpid()}")
    if os.path.exists(env["NMV_BENCH_CAPTURE_MARKER"]):
        os.remove(env["NMV_BENCH_CAPTURE_MARKER"])
    def cmd(extra):
        return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + \
            sys.argv[1:] + extra
    rc = subprocess.call(cmd([]), env=env)
    if rc != 0 and not args.no_graph and os.path.exists(_capture_marker(env)):
        # a rank's hipGraph capture failed: the ranks agreed, left a marker and exited (they never continue eagerly in
        # the same processes).  This process has not touched the GPU: start FRESH ranks without the graph.
        os.remove(_capture_marker(env))
        print("bench.py: hipGraph capture failed in the tensor-parallel ranks; starting fresh ranks with --no-graph",
              file=sys.stderr)
        rc = subprocess.call(cmd(["--no-graph"]), env=env)
    return rc


def main():
    t_main = time.perf_counter()
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
              f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...), "
              f"or run `python bench.py --gpus {args.gpus}` without WORLD_SIZE set", file=sys.stderr)
        sys.exit(2)
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
    arch = {"llama3-8b": dr.LLAMA3_8B, "llama3-70b": dr.LLAMA3_70B, "tiny": dr.TINY, "tiny-70b": dr.TINY_70B}[args.model]
    fit = weights_fit(arch, args.quant, world, args.batch, args.context + args.steps + args.warmup + 8, args.kv_cache_dtype)
    if not fit["fits"]:
        print(f"bench.py: {args.model} {args.quant} at TP={world} needs {fit['total_gb']:.1f} GB per rank "
              f"(weights {fit['weights_gb']:.1f} + KV {fit['kv_gb']:.1f}), more than the 288 GB of an MI355X", file=sys.stderr)
        sys.exit(2)
    quant = {"w4a16": dict(method="gptq_marlin", bits=4, group_size=128),
             "w8a8": dict(method="w8a8", bits=8, group_size=-1), "bf16": None}[args.quant]
    runner = dr.DecodeRunner(arch, dev, torch.bfloat16, quant,
                             dr.CacheConfig(16, args.kv_cache_dtype))

    def max_over_ranks(x: float) -> float:
        if world == 1:
            return x
        vals = [None] * world
        dist.all_gather_object(vals, float(x), group=nd.get_tp_group().cpu_group)
        return max(vals)

    def measure(batch, steps, warmup):
        # every step appends a token and attention is linear in the context, so the timed steps are CENTRED on
        # the nominal context: they start far enough below it that the mean context they attend to is
        # args.context (a 256-step region that started at 512 would measure a mean context of 648)
        start_ctx = max(16, args.context - warmup - steps // 2)
        runner.setup_batch(batch, start_ctx, steps + warmup + 8)
        runner.fill_context()
        # capture() decides graph-or-eager for the whole group (a step that holds collectives of a
        # non-capturable backend is never offered to the graph) and checks the P2P error word
        try:
            graphed = False if args.no_graph else runner.capture()
        except dr.CaptureFailedError as e:
            # agreed by every rank: leave the processes (no eager continuation after a failed capture under TP)
            print(f"bench.py rank {rank}: {e}", file=sys.stderr, flush=True)
            marker = _capture_marker(os.environ)
            if marker and rank == 0:
                open(marker, "w").close()
            os._exit(EXIT_CAPTURE_FAILED)
        for _ in range(warmup):
            runner.decode_step()
        torch.cuda.synchronize(dev)
        if world > 1:
            nd.get_tp_group().barrier()
        torch.cuda.synchronize(dev)
        ctx0 = int(runner.seq_lens[0])       # tokens the first timed step attends to (the new one included)
        t0 = time.perf_counter()
        for _ in range(steps):
            runner.decode_step()
        torch.cuda.synchronize(dev)
        if world > 1:
            nd.get_tp_group().barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        timed_ctx[:] = [ctx0, int(runner.seq_lens[0]) - 1]
        # a P2P all-reduce that gave up waiting for a peer wrote NaN: no number from such a run
        runner.check_collectives()
        return dt, graphed

    timed_ctx = [args.context, args.context]   # context of the first / last step of the last measure() call
    dt, graphed = measure(args.batch, args.steps, args.warmup)
    ctx_first, ctx_last = timed_ctx
    ms_per_step = dt / args.steps * 1e3
    value = args.batch * args.steps / dt

    out = {
        "metric": {"w4a16": "decode tokens/sec, Llama-3-8B w4a16 (GPTQ-marlin g128), bf16 activations",
                   "w8a8": "decode tokens/sec, Llama-3-8B w8a8 (int8 per-channel x dynamic per-token int8)",
                   "bf16": "decode tokens/sec, Llama-3-8B bf16 (unquantised)"}[args.quant],
        "value": round(value, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "timed_region_s": round(dt, 4),
        "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "int8" if args.quant == "w8a8" else "bf16",
        "data": {"w4a16": "synthetic (random-init N(0,0.02) weights quantised to int4 g128, random KV context)",
                 "w8a8": "synthetic (random-init N(0,0.02) weights quantised to int8 per channel, random KV context)",
                 "bf16": "synthetic (random-init N(0,0.02) bf16 weights, random KV context)"}[args.quant],
        "config": {"workload": f"{args.model} {args.quant} decode step, batch {args.batch}, "
                               f"context {args.context} tokens/seq (mean over the timed steps), block 16, "
                               f"kv {args.kv_cache_dtype}",
                   "global_batch": args.batch, "context_len": args.context,
                   # every step appends a token: the KV the timed steps walk grows from / to
                   "context_len_timed_steps": [ctx_first, ctx_last],
                   "parallelism": f"tp{world}", "hip_graph": graphed},
    }
    if dt < 0.5 and not args.no_sweep:
        # the contract times exactly --steps steps; when that region is short, a longer one is measured
        # as well and reported beside it (it never replaces `value`)
        long_steps = 256
        d2, _ = measure(args.batch, long_steps, 4)
        out["sustained"] = {"steps": long_steps, "timed_region_s": round(d2, 4),
                            "value": round(args.batch * long_steps / d2, 1),
                            "ms_per_step": round(d2 / long_steps * 1e3, 4)}
    # measured after the steps (derived tensors built, released ones gone), not the estimate `fit` was made from
    out["config"]["weights_gb_per_rank"] = round(runner.resident_weight_bytes() / 1e9, 2)
    if world > 1:
        # what a reader needs to audit a multi-GPU line from its own content: how many ranks took part, which
        # device each one bound, whether its P2P start-up self-test passed, and the collective library's version
        tp = nd.get_tp_group()
        car = tp.custom_ar
        mine = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.current_device(),
                "device_name": torch.cuda.get_device_name(dev),
                "p2p_selftest": ("passed" if car is not None else "disabled or failed: process-group collectives")}
        seen = [None] * world
        dist.all_gather_object(seen, mine, group=tp.cpu_group)
        out["config"]["ranks_seen"] = len([x for x in seen if x is not None])
        out["config"]["ranks"] = seen
        try:
            out["config"]["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:   # a build without the binding: say so rather than fail the line
            out["config"]["rccl_version"] = f"unavailable ({type(e).__name__})"
        out["config"]["dist_backend"] = backend
    if world > 1:
        # which path carries the row-parallel all-reduces: the P2P kernels over HIP IPC (after their
        # start-up self-test on these devices) or the process group (RCCL)
        tp = nd.get_tp_group()
        car = tp.custom_ar
        if car is not None:
            msg = args.batch * arch.hidden_size * 2
            out["config"]["all_reduce"] = (f"p2p {'two-shot' if car.is_two_shot(msg) else 'one-shot'} over HIP IPC, "
                                           f"fused with residual-add + RMSNorm ({msg} B per call)")
            if os.environ.get("NMV_BENCH_COMPARE_RCCL", "0") == "1" and not args.no_sweep:
                # opt-in (NMV_BENCH_COMPARE_RCCL=1): the same step with the P2P communicator switched off -- every
                # all-reduce / gather through the process group (RCCL) -- so that the scaling curve can be read for
                # both.  Not by default: a fault in that second path would cost the line of the first.
                tp.custom_ar = None
                try:
                    d3, g3 = measure(args.batch, args.steps, args.warmup)
                    out["process_group_path"] = {"backend": backend, "value": round(args.batch * args.steps / d3, 1),
                                                 "ms_per_step": round(d3 / args.steps * 1e3, 4), "hip_graph": g3}
                finally:
                    tp.custom_ar = car
        else:
            out["config"]["all_reduce"] = f"process group ({backend})"
    if rank == 0:
        wb = runner.weight_bytes_per_step()
        kv_elem = 1 if args.kv_cache_dtype.startswith("fp8") else 2
        # KV bytes of a step at the MEAN context of the timed steps (the cache grows by a token per step)
        mean_ctx = (ctx_first + ctx_last) / 2
        kvb = int(2 * mean_ctx * runner.num_kv_heads * arch.head_dim * kv_elem * args.batch * arch.num_hidden_layers)
        out["step_roofline"] = {"weight_bytes": wb, "kv_bytes": kvb, "mean_context": mean_ctx,
                                "hbm_bound_ms": round((wb + kvb) / (HBM_PEAK_GBS * 1e9) * 1e3, 4),
                                "frac_of_hbm_bound": round((wb + kvb) / (HBM_PEAK_GBS * 1e9) / (ms_per_step * 1e-3), 4)}
        if args.quant == "w4a16":
            out["roofline"] = gemm_roofline(runner, args.batch, dev)
    if rank == 0 and world == 1 and not args.no_sweep:
        out["ttft_ms_p50"] = {"prompt_tokens": args.context, "batch": 1,
                              "value": ttft(runner, dev, args.context)}
        if args.quant == "w4a16":
            out["ttft_ms_p50"]["gemm_roofline"] = prefill_gemm_roofline(runner, args.context, dev)
    if rank == 0 and world == 1 and args.e2e:
        out["e2e"] = e2e_latency(runner, dev, prompt_len=args.context)
    if world == 1 and not args.no_sweep:
        sweep = {}
        for b in (1, 8, 32):
            d, _ = measure(b, max(16, args.steps // 2), 4)
            sweep[str(b)] = round(b * max(16, args.steps // 2) / d, 1)
        sweep[str(args.batch)] = round(value, 1)
        out["batch_sweep_tokens_per_s"] = sweep
    if rank == 0 and world == 1 and not args.no_sweep and args.quant == "w4a16" and args.model == "llama3-8b":
        # BASELINE.json's other single-GPU configurations, in the same driver-visible line: config 3 (w8a8 + fp8 KV) and
        # config 1 (bf16), batch args.batch, 64 timed steps each; skipped when the run is already long
        others = {}
        for tag, q, kvd in (("llama3-8b w8a8 + fp8 KV", dict(method="w8a8", bits=8, group_size=-1), "fp8"),
                            ("llama3-8b bf16", None, "auto")):
            if time.perf_counter() - t_main > 75:
                others[tag] = {"skipped": "the run had already taken more than 75 s"}
                continue
            try:
                runner = None
                torch.cuda.empty_cache()
                runner = dr.DecodeRunner(arch, dev, torch.bfloat16, q, dr.CacheConfig(16, kvd))
                d, g3 = measure(args.batch, 64, 4)
                others[tag] = {"value": round(args.batch * 64 / d, 1), "unit": "tokens/s", "ms_per_step": round(d / 64 * 1e3, 4),
                               "steps": 64, "batch": args.batch, "context_len": args.context, "hip_graph": g3}
            except Exception as e:   # never take the headline number down
                others[tag] = {"value": None, "error": repr(e)}
        out["other_configs"] = others
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(arch, args.batch, args.context)
        except Exception as e:  # the baseline must never take the GPU number down with it
            out["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": usable_cores(),
                                   "kind": "port", "sample": f"failed: {e!r}"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        nd.destroy_model_parallel()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
