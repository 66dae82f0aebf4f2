"""GPU parity: rms_norm, fused_add_rms_norm, rotary_embedding, act_and_mul, gelu_* vs the oracle
(which restates the CUDA kernels' rounding) and the reference's golden vectors."""
import os

import numpy as np
import pytest
import torch

import helpers
import oracle

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ULP = {torch.bfloat16: 2**-7, torch.float16: 2**-10}


def close(a, b, dtype, ulps=1.0, atol=1e-6):
    a, b = a.float(), b.float()
    return torch.allclose(a, b, atol=atol, rtol=ulps * ULP[dtype])


@pytest.mark.parametrize("num_tokens", [1, 7, 83])
@pytest.mark.parametrize("hidden", [768, 4096, 5120, 8199])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
@pytest.mark.parametrize("add_residual", [False, True])
def test_rms_norm(gpu_device, num_tokens, hidden, dtype, add_residual):
    from neural_magic_vllm_amd import _custom_ops as ops
    g = torch.Generator().manual_seed(0)
    x = (torch.randn((num_tokens, hidden), generator=g) * (1 / (2 * hidden**0.5))).to(dtype)
    res = torch.randn((num_tokens, hidden), generator=g).to(dtype)
    w = (1 + 0.1 * torch.randn(hidden, generator=g)).to(dtype)
    xd, rd, wd = x.to(gpu_device), res.to(gpu_device), w.to(gpu_device)
    if add_residual:
        r_ref = res.clone()
        ref = oracle.rms_norm(x, w, 1e-6, residual=r_ref)
        ops.fused_add_rms_norm(xd, rd, wd, 1e-6)
        assert torch.equal(rd.cpu(), r_ref)  # x + residual rounded once: exact
        out = xd.cpu()
    else:
        ref = oracle.rms_norm(x, w, 1e-6)
        od = torch.empty_like(xd)
        ops.rms_norm(od, xd, wd, 1e-6)
        out = od.cpu()
    # the variance is summed in a different order, so rsqrt can differ in the last fp32 bit; a
    # flipped intermediate rounding (x*s -> dtype) then costs up to 2 ulps after the weight multiply
    assert close(out, ref, dtype, ulps=2.01)


@pytest.mark.parametrize("is_neox", [True, False])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
@pytest.mark.parametrize("head_size,rot_dim", [(64, 64), (128, 128), (96, 32), (256, 128)])
def test_rotary_embedding(gpu_device, is_neox, dtype, head_size, rot_dim):
    from neural_magic_vllm_amd import _custom_ops as ops
    g = torch.Generator().manual_seed(0)
    nt, nh, nkv, maxpos = 21, 8, 2, 4096
    inv = 1.0 / (10000**(torch.arange(0, rot_dim, 2).float() / rot_dim))
    fr = torch.einsum("i,j->ij", torch.arange(maxpos).float(), inv)
    cache = torch.cat((fr.cos(), fr.sin()), dim=-1).to(dtype)
    pos = torch.randint(0, maxpos, (nt, ), generator=g)
    q = torch.randn((nt, nh * head_size), generator=g).to(dtype)
    k = torch.randn((nt, nkv * head_size), generator=g).to(dtype)
    qd, kd = q.to(gpu_device), k.to(gpu_device)
    ops.rotary_embedding(pos.to(gpu_device), qd, kd, head_size, cache.to(gpu_device), is_neox)
    oracle.rotary_embedding(pos, q, k, head_size, cache, is_neox)
    assert torch.equal(qd.cpu(), q) and torch.equal(kd.cpu(), k)  # same roundings: bit exact


def test_rotary_embedding_strided_qkv(gpu_device):
    """q and k are column slices of the qkv projection output, as in LlamaAttention.forward."""
    from neural_magic_vllm_amd import _custom_ops as ops
    dt, hs, nh, nkv, nt = torch.bfloat16, 128, 8, 2, 9
    g = torch.Generator().manual_seed(1)
    inv = 1.0 / (500000**(torch.arange(0, hs, 2).float() / hs))
    fr = torch.einsum("i,j->ij", torch.arange(1024).float(), inv)
    cache = torch.cat((fr.cos(), fr.sin()), dim=-1).to(dt)
    qkv = torch.randn((nt, (nh + 2 * nkv) * hs), generator=g).to(dt)
    pos = torch.randint(0, 1024, (nt, ), generator=g)
    qkv_d = qkv.to(gpu_device)
    q_d, k_d, _ = qkv_d.split([nh * hs, nkv * hs, nkv * hs], dim=-1)
    ops.rotary_embedding(pos.to(gpu_device), q_d, k_d, hs, cache.to(gpu_device), True)
    q, k, v = qkv.split([nh * hs, nkv * hs, nkv * hs], dim=-1)
    oracle.rotary_embedding(pos, q, k, hs, cache, True)
    assert torch.equal(qkv_d.cpu(), qkv)


def test_batched_rotary_embedding(gpu_device):
    from neural_magic_vllm_amd import _custom_ops as ops
    dt, hs, nt = torch.half, 64, 13
    g = torch.Generator().manual_seed(2)
    inv = 1.0 / (10000**(torch.arange(0, hs, 2).float() / hs))
    fr = torch.einsum("i,j->ij", torch.arange(2048).float(), inv)
    cache = torch.cat((fr.cos(), fr.sin()), dim=-1).to(dt)
    pos = torch.randint(0, 1024, (nt, ), generator=g)
    offs = torch.randint(0, 2, (nt, ), generator=g) * 1024
    q = torch.randn((nt, 4 * hs), generator=g).to(dt)
    k = torch.randn((nt, 4 * hs), generator=g).to(dt)
    qd, kd = q.to(gpu_device), k.to(gpu_device)
    ops.batched_rotary_embedding(pos.to(gpu_device), qd, kd, hs, cache.to(gpu_device), True, hs,
                                 offs.to(gpu_device))
    oracle.rotary_embedding(pos, q, k, hs, cache, True, offsets=offs)
    assert torch.equal(qd.cpu(), q) and torch.equal(kd.cpu(), k)


@pytest.mark.parametrize("act,name", [(0, "silu_and_mul"), (1, "gelu_and_mul"), (2, "gelu_tanh_and_mul")])
@pytest.mark.parametrize("d", [512, 14336, 13])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_act_and_mul(gpu_device, act, name, d, dtype):
    from neural_magic_vllm_amd import _custom_ops as ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn((11, 2 * d), generator=g).to(dtype)
    out = torch.empty((11, d), dtype=dtype, device=gpu_device)
    getattr(ops, name)(out, x.to(gpu_device))
    ref = oracle.act_and_mul(x, act)
    # expf/erff/tanhf differ by an ulp of fp32 between libm and the device: allow 1 ulp of dtype
    assert close(out.cpu(), ref, dtype, ulps=2.01, atol=1e-5)


@pytest.mark.parametrize("name", ["gelu_new", "gelu_fast", "gelu_quick"])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_unary_activations(gpu_device, name, dtype):
    from neural_magic_vllm_amd import _custom_ops as ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn((7, 1000), generator=g).to(dtype)
    out = torch.empty_like(x, device=gpu_device)
    getattr(ops, name)(out, x.to(gpu_device))
    xf = x.float()
    if name == "gelu_quick":
        ref = xf * torch.sigmoid(1.702 * xf)
    elif name == "gelu_new":
        ref = 0.5 * xf * (1 + torch.tanh(0.79788456 * (xf + 0.044715 * xf**3)))
    else:
        ref = 0.5 * xf * (1 + torch.tanh(xf * 0.79788456 * (1 + 0.044715 * xf * xf)))
    # the reference computes gelu_new / gelu_fast with model-dtype intermediates: a few ulps
    assert torch.allclose(out.cpu().float(), ref, atol=2e-2 if dtype == torch.bfloat16 else 3e-3,
                          rtol=4 * ULP[dtype])


def test_glue_golden(gpu_device):
    """against the reference's own CPU kernels (bf16; its CPU build truncates fp32->bf16, so 1 ulp)."""
    from neural_magic_vllm_amd import _custom_ops as ops
    g = np.load(os.path.join(GOLD, "glue_bf16.npz"))
    dt = torch.bfloat16
    T = lambda k: helpers.from_np(g[k], dt)  # noqa: E731
    D = lambda k: T(k).to(gpu_device)  # noqa: E731
    out = torch.empty_like(D("x"))
    ops.rms_norm(out, D("x"), D("w"), 1e-5)
    assert torch.allclose(out.cpu().float(), T("rms").float(), atol=2e-2, rtol=1.6e-2)
    x, r = D("x"), D("res")
    ops.fused_add_rms_norm(x, r, D("w"), 1e-5)
    assert torch.allclose(r.cpu().float(), T("fused_res").float(), atol=0, rtol=2**-7)
    assert torch.allclose(x.cpu().float(), T("fused_x").float(), atol=2e-2, rtol=1.6e-2)
    for name, key in (("silu_and_mul", "silu"), ("gelu_and_mul", "gelu"), ("gelu_tanh_and_mul", "gelu_tanh")):
        o = torch.empty((5, 768), dtype=dt, device=gpu_device)
        getattr(ops, name)(o, D("gate_up"))
        assert torch.allclose(o.cpu().float(), T(key).float(), atol=1e-2, rtol=1.6e-2), name
    pos = torch.from_numpy(g["rope_pos"]).to(gpu_device)
    for neox, tag in ((True, "neox"), (False, "gptj")):
        q, k = D("rope_q"), D("rope_k")
        ops.rotary_embedding(pos, q, k, 128, D("rope_cache"), neox)
        assert torch.allclose(q.cpu().float(), T("rope_q_" + tag).float(), atol=2e-2, rtol=1.6e-2)
        assert torch.allclose(k.cpu().float(), T("rope_k_" + tag).float(), atol=2e-2, rtol=1.6e-2)


# ----------------------------------------------------------------------------- fused launches
# Each fused op must be bit-identical to the sequence of reference ops it replaces (which are
# themselves pinned against the oracle above and in test_gpu_cache.py / test_gpu_w8a8.py).
@pytest.mark.parametrize("is_neox", [True, False])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
@pytest.mark.parametrize("kv_cache_dtype", ["auto", "fp8"])
@pytest.mark.parametrize("heads,kv_heads,head_size,rot_dim", [(32, 8, 128, 128), (8, 8, 64, 64), (4, 1, 128, 64),
                                                             (16, 2, 256, 128)])
def test_rotary_embedding_and_cache_matches_separate_ops(gpu_device, is_neox, dtype, kv_cache_dtype, heads,
                                                         kv_heads, head_size, rot_dim):
    from neural_magic_vllm_amd import _custom_ops as ops
    d = gpu_device
    g = torch.Generator().manual_seed(0)
    num_tokens, block_size, num_blocks, max_pos = 19, 16, 11, 4096
    qkv = torch.randn((num_tokens, (heads + 2 * kv_heads) * head_size), generator=g).to(dtype).to(d)
    cos_sin = torch.randn((max_pos, rot_dim), generator=g).to(dtype).to(d)
    positions = torch.randint(0, max_pos, (num_tokens, ), generator=g).to(d)
    slots = torch.randperm(num_blocks * block_size, generator=g)[:num_tokens].to(d)
    slots[3] = -1  # padding token: rotated, not cached
    cdt = torch.uint8 if kv_cache_dtype == "fp8" else dtype
    x = 16 // torch.tensor([], dtype=cdt).element_size()
    kv_scale = 0.5 if kv_cache_dtype == "fp8" else 1.0

    def caches():
        gen = torch.Generator().manual_seed(1)
        kc = torch.randint(0, 100, (num_blocks, kv_heads, head_size // x, block_size, x), generator=gen)
        vc = torch.randint(0, 100, (num_blocks, kv_heads, head_size, block_size), generator=gen)
        return kc.to(cdt).to(d), vc.to(cdt).to(d)

    def split(t):
        return t.split([heads * head_size, kv_heads * head_size, kv_heads * head_size], dim=-1)

    # the separate ops on strided slices of qkv, as the model calls them
    ref_qkv = qkv.clone()
    q, k, v = split(ref_qkv)
    ref_kc, ref_vc = caches()
    ops.rotary_embedding(positions, q, k, head_size, cos_sin, is_neox)
    ops.reshape_and_cache(k.view(num_tokens, kv_heads, head_size), v.view(num_tokens, kv_heads, head_size),
                          ref_kc, ref_vc, slots, kv_cache_dtype, kv_scale)
    got_qkv = qkv.clone()
    q, k, v = split(got_qkv)
    kc, vc = caches()
    ops.rotary_embedding_and_cache(positions, q, k, v, head_size, cos_sin, is_neox, kc, vc, slots,
                                   kv_cache_dtype, kv_scale)
    assert torch.equal(got_qkv.view(torch.int16), ref_qkv.view(torch.int16))
    assert torch.equal(kc.view(torch.uint8), ref_kc.view(torch.uint8))
    assert torch.equal(vc.view(torch.uint8), ref_vc.view(torch.uint8))
    assert not torch.equal(got_qkv.view(torch.int16), qkv.view(torch.int16))


@pytest.mark.parametrize("num_tokens", [1, 7, 83])
@pytest.mark.parametrize("hidden", [768, 4096, 5120, 8192])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
@pytest.mark.parametrize("add_residual", [False, True])
def test_rms_norm_dynamic_int8_quant_matches_separate_ops(gpu_device, num_tokens, hidden, dtype, add_residual):
    from neural_magic_vllm_amd import _custom_ops as ops
    g = torch.Generator().manual_seed(0)
    d = gpu_device
    x = torch.randn((num_tokens, hidden), generator=g).to(dtype).to(d)
    res = torch.randn((num_tokens, hidden), generator=g).to(dtype).to(d)
    w = (1 + 0.1 * torch.randn((hidden, ), generator=g)).to(dtype).to(d)
    if add_residual:
        normed, ref_res = x.clone(), res.clone()
        ops.fused_add_rms_norm(normed, ref_res, w, 1e-5)
    else:
        normed = torch.empty_like(x)
        ops.rms_norm(normed, x, w, 1e-5)
    ref_q, ref_s = ops.scaled_int8_quant(normed)
    x_in, got_res = x.clone(), res.clone()
    q, s = ops.rms_norm_dynamic_int8_quant(x_in, got_res if add_residual else None, w, 1e-5)
    assert q.dtype == torch.int8 and s.shape == (num_tokens, 1)
    assert torch.equal(q, ref_q) and torch.equal(s, ref_s)
    assert torch.equal(x_in, x)  # the input is left untouched
    if add_residual:
        assert torch.equal(got_res, ref_res)


@pytest.mark.parametrize("num_tokens", [1, 64])
@pytest.mark.parametrize("d_", [512, 14336, 11008, 28672])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_silu_and_mul_dynamic_int8_quant_matches_separate_ops(gpu_device, num_tokens, d_, dtype):
    from neural_magic_vllm_amd import _custom_ops as ops
    g = torch.Generator().manual_seed(0)
    x = (torch.randn((num_tokens, 2 * d_), generator=g) * 2).to(dtype).to(gpu_device)
    act = torch.empty((num_tokens, d_), dtype=dtype, device=gpu_device)
    ops.silu_and_mul(act, x)
    ref_q, ref_s = ops.scaled_int8_quant(act)
    q, s = ops.silu_and_mul_dynamic_int8_quant(x)
    assert torch.equal(q, ref_q) and torch.equal(s, ref_s)


@pytest.mark.parametrize("b,v,stride", [(1, 128256, 128256), (64, 128256, 128256), (5, 2048, 2048),
                                       (3, 1000, 1024), (7, 31, 40)])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_greedy_sample_matches_argmax_with_ties(gpu_device, b, v, stride, dtype):
    from neural_magic_vllm_amd import _custom_ops as ops
    g = torch.Generator().manual_seed(0)
    full = torch.randn((b, stride), generator=g).to(dtype)
    full[:, v:] = 100.0                 # padding columns beyond the vocabulary must be ignored
    logits = full.to(gpu_device)[:, :v]
    got = ops.greedy_sample_advance(logits).cpu()
    # bf16 / fp16 logits have many exact ties: the lowest index wins
    ref = torch.stack([(row == row.max()).nonzero()[0, 0] for row in full[:, :v].float()])
    assert got.dtype == torch.int64 and torch.equal(got, ref)
    flat = torch.zeros((b, stride), dtype=dtype)
    flat[:, min(17, v - 1)] = 1.0
    flat[:, v - 1] = 1.0
    assert torch.equal(ops.greedy_sample_advance(flat.to(gpu_device)[:, :v]).cpu(),
                       torch.full((b, ), min(17, v - 1), dtype=torch.int64))


def test_greedy_sample_advances_the_decode_state(gpu_device):
    """input_ids / positions / seq_lens / slot_mapping after the fused tail == the reference's host
    bookkeeping (model_runner.py:572-580)"""
    from neural_magic_vllm_amd import _custom_ops as ops
    d = gpu_device
    g = torch.Generator().manual_seed(1)
    b, v, bs, nblk = 6, 512, 16, 9
    logits = torch.randn((b, v), generator=g).to(torch.bfloat16).to(d)
    bt = torch.randperm(b * nblk, generator=g).view(b, nblk).to(torch.int32).to(d)
    positions = torch.tensor([0, 14, 15, 16, 31, 100], dtype=torch.int64, device=d)
    seq_lens = (positions + 1).to(torch.int32)
    input_ids = torch.zeros(b, dtype=torch.int64, device=d)
    slots = torch.zeros(b, dtype=torch.int64, device=d)
    pos0 = positions.clone()
    tok = ops.greedy_sample_advance(logits, input_ids, positions, seq_lens, slots, bt, bs)
    assert torch.equal(tok, logits.float().argmax(-1)) and torch.equal(input_ids, tok)
    assert torch.equal(positions, pos0 + 1) and torch.equal(seq_lens.long(), pos0 + 2)
    exp = torch.tensor([int(bt[i, int(positions[i]) // bs]) * bs + int(positions[i]) % bs for i in range(b)])
    assert torch.equal(slots.cpu(), exp)
