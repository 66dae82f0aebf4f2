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
