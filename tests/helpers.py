"""Deterministic synthetic inputs shared by the golden-vector generator and the tests.

Everything is generated with the CPU RNG from explicit seeds (the reference's own tests draw on
the device RNG, which is not reproducible across vendors -- SURVEY.md section 8c), so the same
inputs can be rebuilt here, in tools/make_golden.py and on the GPU box.
"""
import hashlib
import random
from typing import List, Optional, Tuple

import torch

PARTITION_SIZE = 512


def tensor_sha(*tensors: torch.Tensor) -> str:
    h = hashlib.sha256()
    for t in tensors:
        t = t.detach().cpu().contiguous()
        if t.dtype in (torch.bfloat16, torch.float16):
            t = t.view(torch.int16)
        h.update(t.numpy().tobytes())
    return h.hexdigest()[:16]


def to_np(t: torch.Tensor):
    """numpy view that survives bf16 (stored as int16 bit patterns)."""
    t = t.detach().cpu().contiguous()
    if t.dtype in (torch.bfloat16, torch.float16):
        return t.view(torch.int16).numpy()
    return t.numpy()


def from_np(a, dtype: torch.dtype) -> torch.Tensor:
    t = torch.from_numpy(a.copy())
    if dtype in (torch.bfloat16, torch.float16):
        return t.view(dtype)
    return t.to(dtype)


def kv_cache_shapes(num_blocks, block_size, num_kv_heads, head_size, elem_size):
    x = 16 // elem_size
    return ((num_blocks, num_kv_heads, head_size // x, block_size, x),
            (num_blocks, num_kv_heads, head_size, block_size))


def make_paged_attention_inputs(seed: int, num_seqs: int, num_heads: Tuple[int, int],
                                head_size: int, block_size: int, dtype: torch.dtype,
                                seq_lens: Optional[List[int]] = None, max_seq_len: int = 1500,
                                num_blocks: int = 512, use_alibi: bool = False,
                                kv_cache_dtype: str = "auto"):
    """The recipe of tests/kernels/test_attention.py:146-186 of the reference (uniform(-scale,
    scale) q/K/V with scale = head_size**-0.5, random block tables), on the CPU generator."""
    g = torch.Generator().manual_seed(seed)
    rnd = random.Random(seed)
    nq, nkv = num_heads
    scale = float(head_size**-0.5)
    query = (torch.rand((num_seqs, nq, head_size), generator=g) * 2 - 1).mul_(scale).to(dtype)
    if seq_lens is None:
        seq_lens = [rnd.randint(1, max_seq_len) for _ in range(num_seqs)]
        seq_lens[-1] = max_seq_len
    max_len = max(seq_lens)
    max_blocks = (max_len + block_size - 1) // block_size
    block_tables = torch.tensor(
        [[rnd.randint(0, num_blocks - 1) for _ in range(max_blocks)] for _ in range(num_seqs)],
        dtype=torch.int32)
    kshape, vshape = kv_cache_shapes(num_blocks, block_size, nkv, head_size,
                                     1 if kv_cache_dtype != "auto" else torch.empty(0, dtype=dtype).element_size())
    kf = (torch.rand(kshape, generator=g) * 2 - 1).mul_(scale)
    vf = (torch.rand(vshape, generator=g) * 2 - 1).mul_(scale)
    if kv_cache_dtype == "auto":
        key_cache, value_cache = kf.to(dtype), vf.to(dtype)
    else:  # fp8 e4m3fn bytes (the reference fills them through convert_fp8, vllm/utils.py:438-455)
        key_cache = kf.half().float().to(torch.float8_e4m3fn).view(torch.uint8)
        value_cache = vf.half().float().to(torch.float8_e4m3fn).view(torch.uint8)
    alibi = torch.randn(nq, generator=g, dtype=torch.float32) if use_alibi else None
    return dict(query=query, key_cache=key_cache, value_cache=value_cache,
                block_tables=block_tables, seq_lens=torch.tensor(seq_lens, dtype=torch.int32),
                scale=scale, alibi_slopes=alibi, max_seq_len=max_len, num_kv_heads=nkv,
                block_size=block_size)


def blocksparse_mask(seq_len: int, num_heads: int, num_kv_heads: int, tp_rank: int, local_blocks: int,
                     vert_stride: int, block_size: int, head_sliding_step: int) -> torch.Tensor:
    """[heads, seq_len] additive mask (0 / -inf) of block-sparse attention for the LAST token of a sequence, as the
    reference's test builds it (tests/kernels/test_blocksparse_attention.py:117-139; the kernel:
    attention_kernels.cu:209-251): block kb of `block_size` tokens is attended when it is one of the last
    `local_blocks` blocks or when kb + head offset is a multiple of `vert_stride`"""
    qb = (seq_len - 1) // block_size
    mask = torch.full((num_heads, seq_len), float("-inf"))
    for h in range(num_heads):
        if head_sliding_step >= 0:
            off = (tp_rank * num_heads + h) * head_sliding_step + 1
        else:
            off = (tp_rank * num_kv_heads + h // (num_heads // num_kv_heads)) * (-head_sliding_step) + 1
        for kb in range(qb + 1):
            if qb - kb < local_blocks or (kb + off) % vert_stride == 0:
                mask[h, kb * block_size:min((kb + 1) * block_size, seq_len)] = 0
    return mask


def ref_paged_attention_torch(inp, kv_scale: float = 1.0, blocksparse=None) -> torch.Tensor:
    """Plain-torch fp32 gather + softmax (the reference test's own checker,
    tests/kernels/test_attention.py:47-116), used to cross-check the C oracle.  blocksparse = dict(tp_rank,
    local_blocks, vert_stride, block_size, head_sliding_step) adds the block-sparse mask."""
    q = inp["query"].float()
    kc, vc = inp["key_cache"], inp["value_cache"]
    if kc.dtype == torch.uint8:
        kc = kc.view(torch.float8_e4m3fn).float() * kv_scale
        vc = vc.view(torch.float8_e4m3fn).float() * kv_scale
    else:
        kc, vc = kc.float(), vc.float()
    ns, nh, hs = q.shape
    nkv = vc.shape[1]
    bs = vc.shape[3]
    out = torch.empty_like(q)
    for i in range(ns):
        L = int(inp["seq_lens"][i])
        bt = inp["block_tables"][i].long()
        tok = torch.arange(L)
        blk, off = bt[tok // bs], tok % bs
        k = kc[blk, :, :, off, :].reshape(L, nkv, hs)  # [L, kvh, D/x, x] -> [L, kvh, D]
        v = vc[blk, :, :, off]  # [L, kvh, D]
        k = k.repeat_interleave(nh // nkv, dim=1)
        v = v.repeat_interleave(nh // nkv, dim=1)
        att = inp["scale"] * torch.einsum("hd,lhd->hl", q[i], k)
        if inp["alibi_slopes"] is not None:
            att = att + inp["alibi_slopes"].view(-1, 1) * (tok - L + 1).float().view(1, -1)
        if blocksparse is not None and blocksparse["vert_stride"] > 1:
            att = att + blocksparse_mask(L, nh, nkv, **blocksparse)
        att = torch.softmax(att, dim=-1)
        out[i] = torch.einsum("hl,lhd->hd", att, v)
    return out


def make_reshape_and_cache_inputs(seed, num_tokens, num_heads, head_size, block_size, num_blocks,
                                  dtype):
    """tests/kernels/test_cache.py:124-160 of the reference: qkv [T,3,H,D], random distinct slots."""
    g = torch.Generator().manual_seed(seed)
    rnd = random.Random(seed)
    slots = rnd.sample(range(num_blocks * block_size), num_tokens)
    slot_mapping = torch.tensor(slots, dtype=torch.int64)
    qkv = torch.randn((num_tokens, 3, num_heads, head_size), generator=g).to(dtype)
    _, key, value = qkv.unbind(dim=1)
    scale = head_size**-0.5
    kshape, vshape = kv_cache_shapes(num_blocks, block_size, num_heads, head_size,
                                     torch.empty(0, dtype=dtype).element_size())
    key_cache = (torch.rand(kshape, generator=g) * 2 - 1).mul_(scale).to(dtype)
    value_cache = (torch.rand(vshape, generator=g) * 2 - 1).mul_(scale).to(dtype)
    return dict(key=key, value=value, key_cache=key_cache, value_cache=value_cache,
                slot_mapping=slot_mapping)


def make_w4a16_problem(seed, size_m, size_k, size_n, num_bits, group_size, act_order, dtype):
    """tests/kernels/test_marlin_gemm.py:126-179: A ~ N(0,1), W ~ N(0,1) quantised with
    marlin_quantize; expected output a @ w_ref."""
    from oracle import ref_math
    g = torch.Generator().manual_seed(seed)
    a = torch.randn((size_m, size_k), generator=g).to(dtype)
    w = torch.randn((size_k, size_n), generator=g).to(dtype)
    w_ref, mq, ms, g_idx, sort_idx, _ = ref_math.marlin_quantize(w, num_bits, group_size,
                                                                 act_order, g)
    return dict(a=a, w_ref=w_ref.to(dtype), marlin_q_w=mq, marlin_s=ms.to(dtype), g_idx=g_idx,
                sort_indices=sort_idx)


# ---------------------------------------------------------------------------------------------
# Tiny models for the end-to-end fixtures (tools/make_golden_model.py runs the REFERENCE's model code
# on these weights; tests/test_gpu_model_golden.py runs the HIP path on the same tensors).
TINY_LLAMA = dict(hidden_size=512, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=8,
                  num_key_value_heads=2, vocab_size=2048, rms_norm_eps=1e-5, rope_theta=500000.0,
                  max_position_embeddings=8192)
# OPT-125m's head geometry (12 heads x 64, MHA; facebook/opt-125m config) with 2 layers, small vocabulary
TINY_OPT = dict(hidden_size=768, ffn_dim=3072, num_hidden_layers=2, num_attention_heads=12,
                vocab_size=2048, max_position_embeddings=2048)


def is_quantised_linear(name: str) -> bool:
    return name.endswith("_proj.weight")


def tiny_llama_checkpoint(seed: int, dtype: torch.dtype):
    """HF-layout tensors (name -> [out, in]) of a random TINY_LLAMA, CPU generator"""
    a = TINY_LLAMA
    g = torch.Generator().manual_seed(1000 + seed)
    h, inter = a["hidden_size"], a["intermediate_size"]
    hd = h // a["num_attention_heads"]
    nq, nkv = a["num_attention_heads"], a["num_key_value_heads"]

    def w(o, i, std=0.05):
        return (torch.randn((o, i), generator=g) * std).to(dtype)

    def norm(n):
        return (1 + 0.1 * torch.randn(n, generator=g)).to(dtype)

    ck = {"model.embed_tokens.weight": w(a["vocab_size"], h, 0.5)}
    for i in range(a["num_hidden_layers"]):
        p = f"model.layers.{i}."
        ck[p + "self_attn.q_proj.weight"] = w(nq * hd, h)
        ck[p + "self_attn.k_proj.weight"] = w(nkv * hd, h)
        ck[p + "self_attn.v_proj.weight"] = w(nkv * hd, h)
        ck[p + "self_attn.o_proj.weight"] = w(h, nq * hd)
        ck[p + "mlp.gate_proj.weight"] = w(inter, h)
        ck[p + "mlp.up_proj.weight"] = w(inter, h)
        ck[p + "mlp.down_proj.weight"] = w(h, inter)
        ck[p + "input_layernorm.weight"] = norm(h)
        ck[p + "post_attention_layernorm.weight"] = norm(h)
    ck["model.norm.weight"] = norm(h)
    ck["lm_head.weight"] = w(a["vocab_size"], h)
    return ck


def quantize_like_reference(w_kn: torch.Tensor, num_bits: int, group_size: int):
    """(q_w int32 [K, N], s [K/g, N]) of oracle.ref_math.quantize_weights -- pinned bit for bit to the
    reference's quantize_weights by tests/golden/mq_*.npz and again by tools/make_golden_model.py"""
    from oracle import ref_math
    _, q_w, s, _, _ = ref_math.quantize_weights(w_kn, num_bits, group_size, False)
    return q_w, s


def gptq_checkpoint_from_dense(ckpt, num_bits: int = 4, group_size: int = 128):
    """dense HF tensors -> GPTQ checkpoint tensors (qweight / scales / g_idx per quantised linear)"""
    from oracle import ref_math
    out = {}
    for name, w in ckpt.items():
        if not is_quantised_linear(name):
            out[name] = w
            continue
        w_kn = w.t().contiguous()
        k, n = w_kn.shape
        q_w, s = quantize_like_reference(w_kn, num_bits, group_size)
        base = name[:-len(".weight")]
        out[base + ".qweight"] = ref_math.gptq_pack(q_w, num_bits, k, n)
        out[base + ".scales"] = s.to(w.dtype)
        out[base + ".g_idx"] = (torch.arange(k, dtype=torch.int32) // group_size)
    return out


def tiny_opt_checkpoint(seed: int, dtype: torch.dtype):
    """HF-layout tensors of a random TINY_OPT (names of facebook/opt-*: model.decoder....)"""
    a = TINY_OPT
    g = torch.Generator().manual_seed(2000 + seed)
    h, f = a["hidden_size"], a["ffn_dim"]

    def w(o, i, std=0.05):
        return (torch.randn((o, i), generator=g) * std).to(dtype)

    def vec(n, mean=0.0, std=0.05):
        return (mean + std * torch.randn(n, generator=g)).to(dtype)

    d = "model.decoder."
    ck = {d + "embed_tokens.weight": w(a["vocab_size"], h, 0.05),
          d + "embed_positions.weight": w(a["max_position_embeddings"] + 2, h, 0.05)}
    for i in range(a["num_hidden_layers"]):
        p = f"{d}layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            ck[p + f"self_attn.{nm}.weight"] = w(h, h)
            ck[p + f"self_attn.{nm}.bias"] = vec(h)
        ck[p + "self_attn_layer_norm.weight"] = vec(h, 1.0, 0.1)
        ck[p + "self_attn_layer_norm.bias"] = vec(h)
        ck[p + "fc1.weight"] = w(f, h)
        ck[p + "fc1.bias"] = vec(f)
        ck[p + "fc2.weight"] = w(h, f)
        ck[p + "fc2.bias"] = vec(h)
        ck[p + "final_layer_norm.weight"] = vec(h, 1.0, 0.1)
        ck[p + "final_layer_norm.bias"] = vec(h)
    ck[d + "final_layer_norm.weight"] = vec(h, 1.0, 0.1)
    ck[d + "final_layer_norm.bias"] = vec(h)
    return ck
