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
                                     1 if kv_cache_dtype != "auto" else 2)
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


def ref_paged_attention_torch(inp, kv_scale: float = 1.0) -> torch.Tensor:
    """Plain-torch fp32 gather + softmax (the reference test's own checker,
    tests/kernels/test_attention.py:47-116), used to cross-check the C oracle."""
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
    kshape, vshape = kv_cache_shapes(num_blocks, block_size, num_heads, head_size, 2)
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
