"""GPU parity of the prompt (prefill) flash-attention kernel against a plain fp32 reference:
varlen batches, causal mask, GQA, q/k/v as strided slices of one qkv tensor (as the model passes
them).  Recipe and tolerance follow the reference's tests/kernels/test_attention.py:321-387
(ref_multi_query_kv_attention; atol/rtol 1e-3 for fp16, 1e-2-class for bf16)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def ref_attention(q, k, v, seq_lens, scale, alibi_slopes=None, window=None):
    """q [T, H, D], k/v [T, KVH, D] fp32 on the CPU; causal softmax attention per prompt.  alibi_slopes [H]:
    bias slope * (key - query) (the reference's prefix_prefill.py:552-557); window: a query sees the keys fewer
    than `window` positions back (prefix_prefill.py:130-144)"""
    out = torch.empty_like(q)
    rep = q.shape[1] // k.shape[1]
    start = 0
    for L in seq_lens:
        qs = q[start:start + L].transpose(0, 1)                       # [H, L, D]
        ks = k[start:start + L].transpose(0, 1).repeat_interleave(rep, dim=0)
        vs = v[start:start + L].transpose(0, 1).repeat_interleave(rep, dim=0)
        s = (qs @ ks.transpose(1, 2)) * scale
        dist = torch.arange(L)[None, :] - torch.arange(L)[:, None]    # key - query
        if alibi_slopes is not None:
            s = s + alibi_slopes[:, None, None] * dist[None].float()
        mask = dist > 0
        if window:
            mask = mask | (-dist >= window)
        s = s.masked_fill(mask[None], float("-inf"))
        out[start:start + L] = (torch.softmax(s, dim=-1) @ vs).transpose(0, 1)
        start += L
    return out


def _slopes(nq):
    return torch.tensor([2.0 ** (-8.0 * (h + 1) / nq) for h in range(nq)], dtype=torch.float32)


@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
@pytest.mark.parametrize("heads", [(32, 8), (8, 8), (12, 1)])
@pytest.mark.parametrize("head_size", [64, 128])
@pytest.mark.parametrize("seq_lens", [[1], [37, 64, 1, 200], [513], [65, 128, 129],
                                      # >= 2048 64-query tiles with 32 heads: the 128-query workgroup form
                                      [260, 129, 128, 1, 257, 255, 64, 130, 200, 3, 127, 256, 131, 90, 17, 259]])
def test_prefill_attention(gpu_device, dtype, heads, head_size, seq_lens):
    from neural_magic_vllm_amd import _custom_ops as ops
    nq, nkv = heads
    t = sum(seq_lens)
    g = torch.Generator().manual_seed(0)
    qkv = (torch.rand((t, (nq + 2 * nkv) * head_size), generator=g) * 2 - 1).to(dtype)
    scale = head_size**-0.5
    qkv_d = qkv.to(gpu_device)
    q, k, v = qkv_d.split([nq * head_size, nkv * head_size, nkv * head_size], dim=-1)
    q, k, v = q.view(t, nq, head_size), k.view(t, nkv, head_size), v.view(t, nkv, head_size)
    out = torch.full((t, nq, head_size), float("nan"), dtype=dtype, device=gpu_device)
    cu = torch.tensor([0] + torch.tensor(seq_lens).cumsum(0).tolist(), dtype=torch.int32, device=gpu_device)
    ops.prefill_attention(out, q, k, v, cu, max(seq_lens), scale)
    qc, kc, vc = qkv.float().split([nq * head_size, nkv * head_size, nkv * head_size], dim=-1)
    ref = ref_attention(qc.view(t, nq, head_size), kc.view(t, nkv, head_size), vc.view(t, nkv, head_size),
                        seq_lens, scale)
    got = out.float().cpu()
    assert not torch.isnan(got).any()
    tol = 2e-3 if dtype == torch.half else 1.6e-2   # P and the output are rounded to the model dtype
    torch.testing.assert_close(got, ref, atol=tol, rtol=tol)


@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
@pytest.mark.parametrize("head_size", [64, 128])
@pytest.mark.parametrize("alibi,window", [(True, None), (False, 50), (False, 128), (True, 100), (False, 1)])
def test_prefill_attention_alibi_and_sliding_window(gpu_device, dtype, head_size, alibi, window):
    """ALiBi bias and the sliding window of the prompt kernel (rocm_flash_attn.py:244-246 hands both to the
    flash-attention call); windows shorter and longer than a 64-key tile, prompts shorter than the window"""
    from neural_magic_vllm_amd import _custom_ops as ops
    nq, nkv = 8, 2
    seq_lens = [300, 17, 129, 64]
    t = sum(seq_lens)
    g = torch.Generator().manual_seed(3)
    q = (torch.rand((t, nq, head_size), generator=g) * 2 - 1).to(dtype)
    k = (torch.rand((t, nkv, head_size), generator=g) * 2 - 1).to(dtype)
    v = (torch.rand((t, nkv, head_size), generator=g) * 2 - 1).to(dtype)
    slopes = _slopes(nq) if alibi else None
    scale = head_size**-0.5
    out = torch.full((t, nq, head_size), float("nan"), dtype=dtype, device=gpu_device)
    cu = torch.tensor([0] + torch.tensor(seq_lens).cumsum(0).tolist(), dtype=torch.int32, device=gpu_device)
    ops.prefill_attention(out, q.to(gpu_device), k.to(gpu_device), v.to(gpu_device), cu, max(seq_lens), scale,
                          slopes.to(gpu_device) if alibi else None, window)
    ref = ref_attention(q.float(), k.float(), v.float(), seq_lens, scale, slopes, window)
    got = out.float().cpu()
    assert not torch.isnan(got).any()
    tol = 2e-3 if dtype == torch.half else 1.6e-2
    torch.testing.assert_close(got, ref, atol=tol, rtol=tol)


def test_prefill_attention_rejects_unsupported_head(gpu_device):
    from neural_magic_vllm_amd import _custom_ops as ops
    from neural_magic_vllm_amd._lib import NmvError
    q = torch.zeros((4, 2, 80), dtype=torch.half, device=gpu_device)
    cu = torch.tensor([0, 4], dtype=torch.int32, device=gpu_device)
    assert not ops.prefill_attention_supported(80)
    with pytest.raises(NmvError):
        ops.prefill_attention(torch.empty_like(q), q, q, q, cu, 4, 1.0)


@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
@pytest.mark.parametrize("heads", [(32, 8), (4, 4)])
@pytest.mark.parametrize("head_size", [64, 128])
@pytest.mark.parametrize("block_size", [8, 16, 32])
@pytest.mark.parametrize("alibi,window", [(False, None), (True, None), (False, 40), (True, 200)])
def test_prefix_prefill_attention(gpu_device, dtype, heads, head_size, block_size, alibi, window):
    """prefix-enabled prefill: the last q_len tokens of every sequence attend to the whole sequence
    through the paged cache (written with reshape_and_cache, random block tables); the reference is
    the full causal attention restricted to those rows; with ALiBi slopes and / or a sliding window
    (PagedAttention.forward_prefix's last two arguments, paged_attn.py:196-197)"""
    from neural_magic_vllm_amd import _custom_ops as ops
    import helpers
    nq, nkv = heads
    seq_lens = [70, 33, 257, 16]
    q_lens = [70 - 64, 33, 65, 1]          # context: 64, 0, 192, 15 tokens
    g = torch.Generator().manual_seed(1)
    scale = head_size**-0.5
    total = sum(seq_lens)
    q_all = (torch.rand((total, nq, head_size), generator=g) * 2 - 1).to(dtype)
    k_all = (torch.rand((total, nkv, head_size), generator=g) * 2 - 1).to(dtype)
    v_all = (torch.rand((total, nkv, head_size), generator=g) * 2 - 1).to(dtype)
    nb = 64
    max_blocks = (max(seq_lens) + block_size - 1) // block_size
    perm = torch.randperm(nb * 4, generator=g)[:len(seq_lens) * max_blocks].view(len(seq_lens), max_blocks)
    kshape, vshape = helpers.kv_cache_shapes(nb * 4, block_size, nkv, head_size, 2)
    kc = torch.full(kshape, float("nan"), dtype=dtype, device=gpu_device)   # garbage outside the sequences
    vc = torch.full(vshape, float("nan"), dtype=dtype, device=gpu_device)
    slots, start = [], 0
    for i, L in enumerate(seq_lens):
        pos = torch.arange(L)
        slots.append(perm[i][pos // block_size] * block_size + pos % block_size)
    slots = torch.cat(slots).to(torch.int64).to(gpu_device)
    ops.reshape_and_cache(k_all.to(gpu_device), v_all.to(gpu_device), kc, vc, slots, "auto", 1.0)
    # queries = the last q_len tokens of every sequence
    q_new, start = [], 0
    for L, ql in zip(seq_lens, q_lens):
        q_new.append(q_all[start + L - ql:start + L])
        start += L
    q_new = torch.cat(q_new).to(gpu_device)
    out = torch.full_like(q_new, float("nan"))
    qsl = torch.tensor([0] + torch.tensor(q_lens).cumsum(0).tolist(), dtype=torch.int32, device=gpu_device)
    sl = torch.tensor(seq_lens, dtype=torch.int32, device=gpu_device)
    cl = torch.tensor([L - ql for L, ql in zip(seq_lens, q_lens)], dtype=torch.int32, device=gpu_device)
    slopes = _slopes(nq) if alibi else None
    ops.prefix_prefill_attention(out, q_new, kc, vc, perm.to(torch.int32).to(gpu_device), qsl, sl, cl,
                                 max(q_lens), scale, slopes.to(gpu_device) if alibi else None, window)
    ref_full = ref_attention(q_all.float(), k_all.float(), v_all.float(), seq_lens, scale, slopes, window)
    ref, start = [], 0
    for L, ql in zip(seq_lens, q_lens):
        ref.append(ref_full[start + L - ql:start + L])
        start += L
    ref = torch.cat(ref)
    got = out.float().cpu()
    assert not torch.isnan(got).any()
    tol = 2e-3 if dtype == torch.half else 1.6e-2
    torch.testing.assert_close(got, ref, atol=tol, rtol=tol)
