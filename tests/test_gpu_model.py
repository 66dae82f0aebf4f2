"""GPU end to end: a small Llama (w4a16 GPTQ-marlin and bf16) through the whole HIP path --
prefill, KV-cache write, paged-attention decode, W4A16 GEMMs, glue ops, hipGraph replay --
against a plain fp32 CPU reference on the same weights and tokens.
Stated tolerance: mean|dlogit| / mean|logit| < 3e-2 (bf16 activations through 2 layers), and
token-for-token greedy equality wherever the reference's top-2 margin exceeds that error."""
import pytest
import torch

from ref_llama import RefLlama

pytestmark = pytest.mark.gpu


def build(quant, gpu_device, arch=None):
    from neural_magic_vllm_amd.worker import decode_runner as dr
    arch = arch or dr.TINY
    weights = list(dr.synthetic_llama_weights(arch, torch.bfloat16, "cpu", quant, seed=0))
    runner = dr.DecodeRunner(arch, gpu_device, torch.bfloat16, quant, dr.CacheConfig(16, "auto"),
                             weights=[(n, t.clone()) for n, t in weights])
    return arch, dict(weights), runner


@pytest.mark.parametrize("quant", [dict(method="gptq_marlin", bits=4, group_size=128), None,
                                   dict(method="w8a8", bits=8, group_size=-1)],
                         ids=["w4a16", "bf16", "w8a8"])
@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "hipgraph"])
def test_tiny_llama_prefill_then_decode(gpu_device, quant, use_graph):
    arch, weights, runner = build(quant, gpu_device)
    # W8A8 (BASELINE.json configs[3]): the reference applies the same dynamic per-token int8
    # rounding to the activations, weights are exact int8 x fp32 scale
    ref = RefLlama(arch, weights, act_int8=bool(quant) and quant.get("method") == "w8a8")
    batch, prompt_len, new_tokens = 3, 37, 6
    runner.setup_batch(batch, prompt_len, new_tokens + 4)
    g = torch.Generator().manual_seed(0)
    prompts = torch.randint(0, arch.vocab_size, (batch, prompt_len), generator=g)

    # ---- prefill through the HIP path (writes the KV cache) ----
    import neural_magic_vllm_amd.worker.decode_runner as dr
    bs = 16
    dev = gpu_device
    ids = prompts.reshape(-1).to(dev)
    pos = torch.arange(prompt_len, device=dev).repeat(batch)
    blk = torch.gather(runner.block_tables.long(), 1, (torch.arange(prompt_len, device=dev) // bs).expand(batch, -1))
    slots = (blk * bs + (torch.arange(prompt_len, device=dev) % bs)).view(-1)
    from neural_magic_vllm_amd.attention.backends.rocm_hip_attn import ROCmHipAttentionMetadata
    cu = torch.arange(0, (batch + 1) * prompt_len, prompt_len, dtype=torch.int32, device=dev)
    md = ROCmHipAttentionMetadata(
        num_prefills=batch, num_prefill_tokens=batch * prompt_len, num_decode_tokens=0,
        slot_mapping=slots, seq_lens=[prompt_len] * batch,
        seq_lens_tensor=torch.full((batch, ), prompt_len, dtype=torch.int32, device=dev),
        max_query_len=prompt_len, max_prefill_seq_len=prompt_len, max_decode_seq_len=0,
        query_start_loc=cu, seq_start_loc=cu,
        context_lens_tensor=torch.zeros(batch, dtype=torch.int32, device=dev),
        block_tables=runner.block_tables[:, :0], use_cuda_graph=False)
    with torch.inference_mode():
        hidden = runner.model(ids, pos, runner.kv_caches, md)
        logits = runner.model.compute_logits(hidden).float().cpu().view(batch, prompt_len, -1)
    ref_logits = torch.stack([ref.forward(prompts[b]) for b in range(batch)])
    rel = ((logits - ref_logits).abs().mean() / ref_logits.abs().mean()).item()
    assert rel < 3e-2, f"prefill logits rel err {rel}"

    # ---- decode: feed the reference's greedy tokens, compare logits step by step ----
    seqs = [prompts[b].tolist() for b in range(batch)]
    nxt = ref_logits[:, -1].argmax(-1)
    runner.input_ids.copy_(nxt.to(dev))
    if use_graph:
        assert runner.capture()
        runner.input_ids.copy_(nxt.to(dev))
    for step in range(new_tokens):
        for b in range(batch):
            seqs[b].append(int(nxt[b]))
        ref_step = torch.stack([ref.forward(torch.tensor(seqs[b]))[-1] for b in range(batch)])
        hip_next = runner.decode_step().cpu()
        ref_next = ref_step.argmax(-1)
        top2 = ref_step.topk(2, dim=-1).values
        margin = (top2[:, 0] - top2[:, 1])
        tol = 3e-2 * ref_step.abs().mean()
        for b in range(batch):
            if margin[b] > 2 * tol:
                assert int(hip_next[b]) == int(ref_next[b]), f"step {step} seq {b}"
        # keep both sides on the reference's token stream
        nxt = ref_next
        runner.input_ids.copy_(nxt.to(dev))


def test_kv_cache_slots_bit_exact(gpu_device):
    """slot = block_table[pos // bs] * bs + pos % bs (model_runner.py:572-580): after a decode
    step the cache holds the new K/V exactly where the reference's indexing puts them."""
    arch, weights, runner = build(None, gpu_device)
    runner.setup_batch(4, 21, 8)
    runner.fill_context()
    before = [kv.clone() for kv in runner.kv_caches]
    pos = runner.positions.clone()
    slots = runner.slot_mapping.clone().cpu()
    bt = runner.block_tables.cpu()
    exp = torch.tensor([int(bt[i, int(pos[i]) // 16]) * 16 + int(pos[i]) % 16 for i in range(4)])
    assert torch.equal(slots, exp)
    runner.decode_step()
    torch.cuda.synchronize()
    kv0, old0 = runner.kv_caches[0], before[0]
    nkv, hd = runner.num_kv_heads, arch.head_dim
    kc = kv0[0].view(-1, nkv, hd // 8, 16, 8)
    ko = old0[0].view(-1, nkv, hd // 8, 16, 8)
    changed = (kc != ko).any(dim=(1, 2, 4)).nonzero().cpu()  # (block, offset) pairs that changed
    got = sorted((int(b) * 16 + int(o)) for b, o in changed)
    assert got == sorted(exp.tolist())
    # the runner advanced its state on device
    assert torch.equal(runner.positions.cpu(), pos.cpu() + 1)


def test_tiny_llama_chunked_prefill_matches_single_shot(gpu_device):
    """prefix-enabled prefill end to end: a prompt fed in two chunks (the second one attends to the
    first through the paged cache, PagedAttention.forward_prefix) must give the logits of the
    single-shot prefill"""
    from neural_magic_vllm_amd.attention.backends.rocm_hip_attn import ROCmHipAttentionMetadata
    arch, weights, runner = build(None, gpu_device)
    dev, bs = gpu_device, 16
    batch, prompt_len, first = 2, 50, 32
    runner.setup_batch(batch, prompt_len, 8)
    g = torch.Generator().manual_seed(3)
    prompts = torch.randint(0, arch.vocab_size, (batch, prompt_len), generator=g).to(dev)

    def meta(lo, hi):
        n = hi - lo
        pos = torch.arange(lo, hi, device=dev)
        blk = torch.gather(runner.block_tables.long(), 1, (pos // bs).expand(batch, -1))
        slots = (blk * bs + pos % bs).view(-1)
        cu = torch.arange(0, (batch + 1) * n, n, dtype=torch.int32, device=dev)
        return ROCmHipAttentionMetadata(
            num_prefills=batch, num_prefill_tokens=batch * n, num_decode_tokens=0, slot_mapping=slots,
            seq_lens=[hi] * batch, seq_lens_tensor=torch.full((batch, ), hi, dtype=torch.int32, device=dev),
            max_query_len=n, max_prefill_seq_len=hi, max_decode_seq_len=0, query_start_loc=cu,
            seq_start_loc=torch.arange(0, (batch + 1) * hi, hi, dtype=torch.int32, device=dev),
            context_lens_tensor=torch.full((batch, ), lo, dtype=torch.int32, device=dev),
            block_tables=runner.block_tables if lo > 0 else runner.block_tables[:, :0],
            use_cuda_graph=False), pos.repeat(batch)

    with torch.inference_mode():
        md, pos = meta(0, prompt_len)
        full = runner.model(prompts.reshape(-1), pos, runner.kv_caches, md).view(batch, prompt_len, -1)
        for kc in runner.kv_caches:   # forget the cache: the chunked run must rebuild it
            kc.zero_()
        md, pos = meta(0, first)
        runner.model(prompts[:, :first].reshape(-1), pos, runner.kv_caches, md)
        md, pos = meta(first, prompt_len)
        part = runner.model(prompts[:, first:].reshape(-1), pos, runner.kv_caches, md)
        part = part.view(batch, prompt_len - first, -1)
    a, b = full[:, first:].float(), part.float()
    rel = ((a - b).abs().mean() / a.abs().mean()).item()
    assert rel < 1e-2, rel


@pytest.mark.parametrize("quant", [dict(method="gptq_marlin", bits=4, group_size=128),
                                   dict(method="w8a8", bits=8, group_size=-1)], ids=["w4a16", "w8a8"])
@pytest.mark.parametrize("wide_mlp", [False, True], ids=["tiny", "wide_mlp"])
def test_fused_glue_is_bit_identical(gpu_device, monkeypatch, quant, wide_mlp):
    """the one-launch forms (rotary + cache write; W8A8: norm / silu_and_mul + dynamic int8
    quantisation; W4A16 with a wide MLP: silu_and_mul in the gate_up GEMM's epilogue) against the
    reference's op-by-op sequence on the same weights: same tokens, same KV cache, bit for bit"""
    import dataclasses
    from neural_magic_vllm_amd.worker import decode_runner as dr
    arch = dataclasses.replace(dr.TINY, intermediate_size=8192, num_hidden_layers=1) if wide_mlp else None
    outs = {}
    for fused in (True, False):
        monkeypatch.setenv("NMV_FUSED_GLUE", "1" if fused else "0")
        _, _, runner = build(quant, gpu_device, arch)
        mods = [m for m in runner.model.modules() if hasattr(m, "fused_glue")]
        assert mods and all(m.fused_glue == fused for m in mods)
        runner.setup_batch(5, 40, 8)
        runner.fill_context()
        toks = [runner.decode_step().clone() for _ in range(4)]
        torch.cuda.synchronize()
        outs[fused] = (torch.stack(toks).cpu(), [kv.clone() for kv in runner.kv_caches])
        if fused and wide_mlp and quant["method"] == "gptq_marlin":
            assert all(getattr(l.mlp.gate_up_proj, "gate_up_interleaved", False)
                       for l in runner.model.model.layers)
    assert torch.equal(outs[True][0], outs[False][0])
    for a, b in zip(outs[True][1], outs[False][1]):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))


@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "hipgraph"])
def test_weight_prefetch_hint_changes_no_result(gpu_device, monkeypatch, use_graph):
    """the Infinity-Cache prefetch of the next layer's weights (models/llama.py: LlamaModel.forward, a side stream of
    load-only launches) is a hint: same tokens, same KV cache with and without it, eagerly and as a branch of the graph"""
    import dataclasses
    from neural_magic_vllm_amd.worker import decode_runner as dr
    from neural_magic_vllm_amd import _custom_ops as ops
    arch = dataclasses.replace(dr.TINY, num_hidden_layers=3)
    quant = dict(method="gptq_marlin", bits=4, group_size=128)
    outs, calls = {}, {}
    real = ops.prefetch_l3
    for on in (True, False):
        monkeypatch.setenv("NMV_PREFETCH", "1" if on else "0")
        n = [0]

        def counted(t, wgs=0, n=n):
            n[0] += 1
            real(t, wgs)
        monkeypatch.setattr(ops, "prefetch_l3", counted)
        _, _, runner = build(quant, gpu_device, arch)
        runner.setup_batch(5, 40, 8)
        runner.fill_context()
        if use_graph:
            assert runner.capture()
        toks = [runner.decode_step().clone() for _ in range(4)]
        torch.cuda.synchronize()
        outs[on], calls[on] = (torch.stack(toks).cpu(), [kv.clone() for kv in runner.kv_caches]), n[0]
    assert calls[True] > 0 and calls[False] == 0
    assert torch.equal(outs[True][0], outs[False][0])
    for a, b in zip(outs[True][1], outs[False][1]):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))


def test_prefetch_l3_reads_any_tail_and_refuses_misaligned_pointers(gpu_device):
    from neural_magic_vllm_amd import _custom_ops as ops
    for n in (0, 1, 15, 16, 4097, 1 << 20, (1 << 22) + 24):
        t = torch.zeros(n, dtype=torch.uint8, device=gpu_device)
        ops.prefetch_l3(t)
        ops.prefetch_l3(t, 7)
    torch.cuda.synchronize()
    t = torch.zeros(4096, dtype=torch.uint8, device=gpu_device)[4:]
    with pytest.raises(Exception, match="aligned"):
        ops.prefetch_l3(t)


def test_runner_refuses_to_run_past_its_tables(gpu_device):
    """positions index the rotary table and the block tables on the device: the runner must stop on the
    host instead (a context beyond max_position_embeddings or a step beyond max_new_tokens would be an
    out-of-bounds read in a kernel)"""
    arch, _, runner = build(None, gpu_device)
    with pytest.raises(ValueError, match="max_position_embeddings"):
        runner.setup_batch(2, arch.max_position_embeddings - 2, 8)
    runner.setup_batch(2, 20, 3)
    runner.fill_context()
    for _ in range(3):
        runner.decode_step()
    with pytest.raises(RuntimeError, match="max_new_tokens"):
        runner.decode_step()


@pytest.mark.parametrize("quant", [dict(method="gptq_marlin", bits=4, group_size=128), None], ids=["gptq", "bf16"])
def test_from_pretrained_safetensors_round_trip(gpu_device, tmp_path, quant):
    """an HF-layout checkpoint on disk (two safetensors shards, config.json, AutoGPTQ's quantize_config.json)
    through DecodeRunner.from_pretrained decodes the same tokens as the same tensors handed over directly"""
    import json
    from safetensors.torch import save_file
    from neural_magic_vllm_amd.worker import decode_runner as dr
    arch, weights, runner = build(quant, gpu_device)
    names = list(weights)
    half = len(names) // 2
    save_file({n: weights[n].contiguous() for n in names[:half]}, str(tmp_path / "model-00001-of-00002.safetensors"))
    save_file({n: weights[n].contiguous() for n in names[half:]}, str(tmp_path / "model-00002-of-00002.safetensors"))
    (tmp_path / "config.json").write_text(json.dumps(dict(
        architectures=["LlamaForCausalLM"], hidden_size=arch.hidden_size, intermediate_size=arch.intermediate_size,
        num_hidden_layers=arch.num_hidden_layers, num_attention_heads=arch.num_attention_heads,
        num_key_value_heads=arch.num_key_value_heads, vocab_size=arch.vocab_size, rms_norm_eps=arch.rms_norm_eps,
        rope_theta=arch.rope_theta, max_position_embeddings=arch.max_position_embeddings, hidden_act="silu",
        torch_dtype="bfloat16")))
    if quant is not None:
        (tmp_path / "quantize_config.json").write_text(json.dumps(dict(
            bits=4, group_size=128, desc_act=False, sym=True, damp_percent=0.01, true_sequential=True)))
    loaded = dr.DecodeRunner.from_pretrained(str(tmp_path), gpu_device, cache_config=dr.CacheConfig(16, "auto"))
    if quant is not None:
        from neural_magic_vllm_amd.model_executor.layers.quantization.gptq_marlin import GPTQMarlinConfig
        assert isinstance(loaded.model.model.layers[0].self_attn.qkv_proj.quant_method.quant_config, GPTQMarlinConfig)
    outs = []
    for r in (runner, loaded):
        r.setup_batch(3, 30, 8)
        r.fill_context()
        outs.append(torch.stack([r.decode_step().clone() for _ in range(4)]).cpu())
    assert torch.equal(outs[0], outs[1])


def test_full_size_llama3_8b_layer_decode_step(gpu_device):
    """ONE decoder layer at Llama-3-8B's real dimensions (hidden 4096, 32 query / 8 KV heads, MLP 14336; w4a16 g128),
    64 sequences with 512 tokens of context: prompt step + one decode step through the HIP path -- the headline
    configuration's kernels at their real shapes (stream GEMM at M = 64 in the deferred / silu forms, fused rope +
    cache write + paged attention, slab-summing norms) -- against the plain fp32 reference of the same math on the
    same weights and tokens (RefLlama, pinned to the reference's own model code by tests/test_oracle_golden.py).
    Stated tolerance: mean|dlogit| / mean|logit| < 3e-2 (bf16 activations, fp32 reference), greedy token equal
    wherever the reference's top-2 margin exceeds twice that."""
    from neural_magic_vllm_amd.worker import decode_runner as dr
    arch = dr.LlamaArch(4096, 14336, 1, 32, 8, 2048)
    quant = dict(method="gptq_marlin", bits=4, group_size=128)
    weights = list(dr.synthetic_llama_weights(arch, torch.bfloat16, "cpu", quant, seed=0))
    runner = dr.DecodeRunner(arch, gpu_device, torch.bfloat16, quant, dr.CacheConfig(16, "auto"),
                             weights=[(n, t.clone()) for n, t in weights])
    ref = RefLlama(arch, {n: t.to(gpu_device) for n, t in weights})
    batch, ctx = 64, 512
    runner.setup_batch(batch, ctx, 8)
    seed = 3
    g = torch.Generator().manual_seed(seed)       # the ids DecodeRunner.prefill draws for this seed
    prompts = torch.randint(0, arch.vocab_size, (batch * ctx, ), generator=g).view(batch, ctx)
    first = runner.prefill(ctx, seed=seed).view(-1).cpu()
    ref_prompt_last = torch.stack([ref.forward(prompts[b].to(gpu_device))[-1] for b in range(batch)]).float().cpu()
    tol = 3e-2 * ref_prompt_last.abs().mean()
    top2 = ref_prompt_last.topk(2, dim=-1).values
    sure = (top2[:, 0] - top2[:, 1]) > 2 * tol
    assert sure.sum() > batch // 2, "the test needs decisive reference tokens"
    assert torch.equal(first[sure], ref_prompt_last.argmax(-1)[sure])
    # the decode step on the tokens the HIP path drew (context 512 + 1)
    runner.keep_logits = True
    runner.input_ids.copy_(first.to(gpu_device))
    runner.decode_step()
    got = runner.last_logits.float().cpu()
    seqs = torch.cat([prompts, first.view(-1, 1)], dim=1)
    want = torch.stack([ref.forward(seqs[b].to(gpu_device))[-1] for b in range(batch)]).float().cpu()
    rel = ((got - want).abs().mean() / want.abs().mean()).item()
    assert rel < 3e-2, f"decode-step logits of the full-size layer: relative error {rel}"
