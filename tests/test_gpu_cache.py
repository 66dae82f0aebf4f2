"""GPU parity: KV-cache ops vs oracle / golden.  reshape_and_cache and copy_blocks are byte
movement: BIT-EXACT (reference tests/kernels/test_cache.py:201-202)."""
import os
import random

import numpy as np
import pytest
import torch

import helpers
import oracle

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("num_heads", [8])
@pytest.mark.parametrize("head_size", [64, 80, 96, 112, 128, 192, 256])
@pytest.mark.parametrize("block_size", [8, 16, 32])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_reshape_and_cache(gpu_device, num_heads, head_size, block_size, dtype):
    from neural_magic_vllm_amd import _custom_ops as ops
    inp = helpers.make_reshape_and_cache_inputs(0, 42, num_heads, head_size, block_size, 128, dtype)
    kc, vc = inp["key_cache"].to(gpu_device), inp["value_cache"].to(gpu_device)
    # key/value are strided views of qkv, as in the reference test
    ops.reshape_and_cache(inp["key"].to(gpu_device), inp["value"].to(gpu_device), kc, vc,
                          inp["slot_mapping"].to(gpu_device), "auto", 1.0)
    kr, vr = inp["key_cache"].clone(), inp["value_cache"].clone()
    oracle.reshape_and_cache(inp["key"], inp["value"], kr, vr, inp["slot_mapping"])
    assert torch.equal(kc.cpu(), kr) and torch.equal(vc.cpu(), vr)


def test_reshape_and_cache_strided_and_padding(gpu_device):
    from neural_magic_vllm_amd import _custom_ops as ops
    dt = torch.bfloat16
    inp = helpers.make_reshape_and_cache_inputs(1, 33, 8, 128, 16, 64, dt)
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn((33, 3, 8, 128), generator=g).to(dt)
    _, key, value = qkv.unbind(dim=1)  # stride(0) = 3*8*128
    slots = inp["slot_mapping"].clone()
    slots[::5] = -1  # padding tokens are skipped (cache_kernels.cu:166-169)
    qkv_d = qkv.to(gpu_device)
    _, key_d, value_d = qkv_d.unbind(dim=1)
    kc, vc = inp["key_cache"].to(gpu_device), inp["value_cache"].to(gpu_device)
    ops.reshape_and_cache(key_d, value_d, kc, vc, slots.to(gpu_device), "auto", 1.0)
    kr, vr = inp["key_cache"].clone(), inp["value_cache"].clone()
    oracle.reshape_and_cache(key, value, kr, vr, slots)
    assert torch.equal(kc.cpu(), kr) and torch.equal(vc.cpu(), vr)


@pytest.mark.parametrize("name", ["rc_bf16_h8_d128_b16", "rc_bf16_h2_d80_b16"])
def test_reshape_and_cache_golden(gpu_device, name):
    from neural_magic_vllm_amd import _custom_ops as ops
    g = np.load(os.path.join(GOLD, name + ".npz"))
    inp = helpers.make_reshape_and_cache_inputs(int(g["seed"]), int(g["num_tokens"]),
                                                int(g["num_heads"]), int(g["head_size"]),
                                                int(g["block_size"]), int(g["num_blocks"]),
                                                torch.bfloat16)
    kc, vc = inp["key_cache"].to(gpu_device), inp["value_cache"].to(gpu_device)
    ops.reshape_and_cache(inp["key"].to(gpu_device), inp["value"].to(gpu_device), kc, vc,
                          inp["slot_mapping"].to(gpu_device), "auto", 1.0)
    assert helpers.tensor_sha(kc.cpu(), vc.cpu()) == str(g["cache_sha"])


@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
@pytest.mark.parametrize("head_size", [64, 128])
@pytest.mark.parametrize("block_size", [8, 16, 32])
def test_reshape_and_cache_fp8(gpu_device, dtype, head_size, block_size):
    """fp8 bytes must equal the oracle's e4m3fn encoding of x / kv_scale."""
    from neural_magic_vllm_amd import _custom_ops as ops
    nb, nh = 64, 8
    inp = helpers.make_reshape_and_cache_inputs(2, 42, nh, head_size, block_size, nb, dtype)
    kshape, vshape = helpers.kv_cache_shapes(nb, block_size, nh, head_size, 1)
    kc = torch.zeros(kshape, dtype=torch.uint8, device=gpu_device)
    vc = torch.zeros(vshape, dtype=torch.uint8, device=gpu_device)
    ops.reshape_and_cache(inp["key"].to(gpu_device), inp["value"].to(gpu_device), kc, vc,
                          inp["slot_mapping"].to(gpu_device), "fp8", 0.37)
    kr, vr = torch.zeros(kshape, dtype=torch.uint8), torch.zeros(vshape, dtype=torch.uint8)
    oracle.reshape_and_cache(inp["key"], inp["value"], kr, vr, inp["slot_mapping"], "fp8", 0.37)
    assert torch.equal(kc.cpu(), kr) and torch.equal(vc.cpu(), vr)


def test_reshape_and_cache_flash(gpu_device):
    from neural_magic_vllm_amd import _custom_ops as ops
    dt, nt, nh, hs, bs, nb = torch.bfloat16, 42, 8, 128, 16, 64
    g = torch.Generator().manual_seed(0)
    rnd = random.Random(0)
    slots = torch.tensor(rnd.sample(range(nb * bs), nt), dtype=torch.int64)
    qkv = torch.randn((nt, 3, nh, hs), generator=g).to(dt)
    _, key, value = qkv.to(gpu_device).unbind(dim=1)
    kc = torch.randn((nb, bs, nh, hs), generator=g).to(dt)
    vc = torch.randn((nb, bs, nh, hs), generator=g).to(dt)
    kd, vd = kc.to(gpu_device), vc.to(gpu_device)
    ops.reshape_and_cache_flash(key, value, kd, vd, slots.to(gpu_device), "auto")
    blk, off = slots // bs, slots % bs
    kc[blk, off] = qkv[:, 1]
    vc[blk, off] = qkv[:, 2]
    assert torch.equal(kd.cpu(), kc) and torch.equal(vd.cpu(), vc)


@pytest.mark.parametrize("num_layers", [1, 5])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.uint8])
def test_copy_blocks(gpu_device, num_layers, dtype):
    from neural_magic_vllm_amd import _custom_ops as ops
    nb, bs, nh, hs = 128, 16, 8, 128
    rnd = random.Random(0)
    g = torch.Generator().manual_seed(0)
    src = rnd.sample(range(nb), 16)
    rest = list(set(range(nb)) - set(src))
    dst = rnd.sample(rest, 32)
    mapping = [(s, dst[2 * i]) for i, s in enumerate(src)] + [(s, dst[2 * i + 1]) for i, s in enumerate(src)]
    bm = torch.tensor(mapping, dtype=torch.int64)
    esz = 1 if dtype == torch.uint8 else 2
    kshape, vshape = helpers.kv_cache_shapes(nb, bs, nh, hs, esz)
    mk = (lambda shp: torch.randint(0, 255, shp, generator=g, dtype=torch.uint8)) if dtype == torch.uint8 \
        else (lambda shp: torch.randn(shp, generator=g).to(dtype))
    kcs = [mk(kshape) for _ in range(num_layers)]
    vcs = [mk(vshape) for _ in range(num_layers)]
    kd, vd = [t.to(gpu_device) for t in kcs], [t.to(gpu_device) for t in vcs]
    ops.copy_blocks(kd, vd, bm.to(gpu_device))
    for kc, vc, a, b in zip(kcs, vcs, kd, vd):
        oracle.copy_blocks(kc, vc, bm)
        assert torch.equal(a.cpu(), kc) and torch.equal(b.cpu(), vc)


def test_copy_blocks_golden(gpu_device):
    from neural_magic_vllm_amd import _custom_ops as ops
    g = np.load(os.path.join(GOLD, "copy_blocks_bf16.npz"))
    inp = helpers.make_reshape_and_cache_inputs(int(g["seed"]), 4, 4, 64, 16, 32, torch.bfloat16)
    kc, vc = inp["key_cache"].to(gpu_device), inp["value_cache"].to(gpu_device)
    ops.copy_blocks([kc], [vc], torch.from_numpy(g["mapping"]).to(gpu_device))
    assert helpers.tensor_sha(kc.cpu(), vc.cpu()) == str(g["cache_sha"])


@pytest.mark.parametrize("direction", [("cuda", "cuda"), ("cuda", "cpu"), ("cpu", "cuda")])
def test_swap_blocks(gpu_device, direction):
    from neural_magic_vllm_amd import _custom_ops as ops
    nb, bs, nh, hs = 64, 16, 8, 128
    g = torch.Generator().manual_seed(0)
    kshape, _ = helpers.kv_cache_shapes(nb, bs, nh, hs, 2)
    src = torch.randn(kshape, generator=g).to(torch.bfloat16)
    dst = torch.randn(kshape, generator=g).to(torch.bfloat16)
    sd = src.to(gpu_device) if direction[0] == "cuda" else src.clone().pin_memory()
    dd = dst.to(gpu_device) if direction[1] == "cuda" else dst.clone().pin_memory()
    rnd = random.Random(0)
    pairs = list(zip(rnd.sample(range(nb), 20), rnd.sample(range(nb), 20)))
    ops.swap_blocks(sd, dd, torch.tensor(pairs, dtype=torch.int64))
    torch.cuda.synchronize()
    exp = dst.clone()
    for s, d in pairs:
        exp[d] = src[s]
    assert torch.equal(dd.cpu(), exp)


@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_convert_fp8_round_trip(gpu_device, dtype):
    from neural_magic_vllm_amd import _custom_ops as ops
    g = torch.Generator().manual_seed(0)
    x = ((torch.rand((64, 8, 128, 16), generator=g) * 2 - 1) * 3).to(dtype)
    xd = x.to(gpu_device)
    f8 = torch.empty(x.shape, dtype=torch.uint8, device=gpu_device)
    ops.convert_fp8(f8, xd, 0.5, "fp8")
    assert torch.equal(f8.cpu(), oracle.fp8_encode(x.float() / 0.5))
    back = torch.empty_like(xd)
    ops.convert_fp8(back, f8, 0.5, "fp8")
    exp = (oracle.fp8_decode(f8.cpu()) * 0.5).to(dtype)
    assert torch.equal(back.cpu(), exp)
    assert torch.allclose(back.cpu().float(), x.float(), atol=1e-3, rtol=0.07)  # e4m3: 3 mantissa bits


@pytest.mark.parametrize("head_size,block_size", [(128, 16), (80, 32), (64, 8)])
def test_reshape_and_cache_and_convert_fp8_float32(gpu_device, head_size, block_size):
    """float models (cache_kernels.cu:253-278, :339-389 are instantiated for float): K cache chunks of x = 4 floats;
    bit-exact against the oracle's scatter; fp8 cache from float key / value; convert_fp8 float <-> fp8 round trip"""
    from neural_magic_vllm_amd import _custom_ops as ops
    inp = helpers.make_reshape_and_cache_inputs(3, 29, 4, head_size, block_size, 64, torch.float32)
    assert inp["key_cache"].shape[-1] == 4
    slots = inp["slot_mapping"].clone()
    slots[::7] = -1
    kc, vc = inp["key_cache"].to(gpu_device), inp["value_cache"].to(gpu_device)
    ops.reshape_and_cache(inp["key"].to(gpu_device), inp["value"].to(gpu_device), kc, vc, slots.to(gpu_device), "auto", 1.0)
    # the layout of cache_kernels.cu:152-204 restated for x = 4 (the C oracle is built for the 16-bit dtypes)
    kr, vr = inp["key_cache"].clone(), inp["value_cache"].clone()
    for t, slot in enumerate(slots.tolist()):
        if slot < 0:
            continue
        b, off = divmod(slot, block_size)
        kr[b, :, :, off, :] = inp["key"][t].reshape(4, head_size // 4, 4)
        vr[b, :, :, off] = inp["value"][t]
    assert torch.equal(kc.cpu(), kr) and torch.equal(vc.cpu(), vr)
    # float -> fp8 -> float: the round trip is the fp8 rounding of x / scale, times scale
    f8 = torch.empty(kc.shape, dtype=torch.uint8, device=gpu_device)
    ops.convert_fp8(f8, kc, 0.25, "fp8")
    back = torch.empty_like(kc)
    ops.convert_fp8(back, f8, 0.25, "fp8")
    want = (kr / 0.25).to(torch.float8_e4m3fn).float() * 0.25
    assert torch.equal(back.cpu(), want)
