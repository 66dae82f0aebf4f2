"""GPU parity: gptq_marlin_repack (bit-exact) and gptq_marlin_gemm (W4A16 MFMA path) vs the
oracle.  Shapes / tolerance follow the reference's tests/kernels/test_marlin_gemm.py:32-45,
:62-114 (repack bit-equal to marlin_weights), :126-179 (mean|d|/mean|ref| < 0.04)."""
import os

import numpy as np
import pytest
import torch

import helpers
import oracle
from oracle import ref_math

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

K_SIZES = [128, 1024, 640, 1664]
N_SIZES = [64, 256, 448, 1088, 2368]


def hip_gemm(pr, m, n, k, bits, dev, is_k_full=True):
    from neural_magic_vllm_amd import _custom_ops as ops
    ws = torch.zeros(max(n // 64 * 16, 16), dtype=torch.int32, device=dev)
    g_idx = pr["g_idx"].to(dev)
    sort_idx = pr["sort_indices"].to(dev)
    c = ops.gptq_marlin_gemm(pr["a"].to(dev), pr["marlin_q_w"].to(dev), pr["marlin_s"].to(dev),
                             g_idx, sort_idx, ws, bits, m, n, k, is_k_full)
    assert int(ws.abs().sum()) == 0, "workspace must be returned zeroed"
    return c.cpu()


@pytest.mark.parametrize("k", K_SIZES)
@pytest.mark.parametrize("n", N_SIZES)
@pytest.mark.parametrize("bits", [4, 8])
@pytest.mark.parametrize("act_order", [False, True])
def test_marlin_repack(gpu_device, k, n, bits, act_order):
    """GPTQ layout -> Marlin layout, bit-equal to the Python marlin_weights of the reference."""
    from neural_magic_vllm_amd import _custom_ops as ops
    g = torch.Generator().manual_seed(0)
    q_w = torch.randint(0, 2**bits, (k, n), generator=g, dtype=torch.int32)
    perm = torch.randperm(k, generator=g).to(torch.int32) if act_order else torch.empty(0, dtype=torch.int32)
    packed = ref_math.gptq_pack(q_w, bits, k, n)
    out = ops.gptq_marlin_repack(packed.to(gpu_device), perm.to(gpu_device), k, n, bits).cpu()
    src = q_w[perm.long()] if act_order else q_w
    assert torch.equal(out, ref_math.marlin_weights(src, k, n, bits))
    assert torch.equal(out, oracle.gptq_marlin_repack(packed, perm if act_order else None, k, n, bits))


@pytest.mark.parametrize("name", ["mq_k256_n128_b4_g128", "mq_k128_n64_b4_gm1",
                                  "mq_k256_n192_b4_g32", "mq_k128_n128_b8_g64"])
def test_marlin_repack_golden(gpu_device, name):
    from neural_magic_vllm_amd import _custom_ops as ops
    g = np.load(os.path.join(GOLD, name + ".npz"))
    k, n, bits = int(g["size_k"]), int(g["size_n"]), int(g["num_bits"])
    packed = torch.from_numpy(g["gptq_packed"]).to(gpu_device)
    e = torch.empty(0, dtype=torch.int32, device=gpu_device)
    assert np.array_equal(ops.gptq_marlin_repack(packed, e, k, n, bits).cpu().numpy(), g["marlin_q_w"])
    perm = torch.from_numpy(g["perm"]).to(gpu_device)
    assert np.array_equal(ops.gptq_marlin_repack(packed, perm, k, n, bits).cpu().numpy(),
                          g["marlin_q_w_perm"])


@pytest.mark.parametrize("k", K_SIZES)
@pytest.mark.parametrize("n", [64, 448, 1088])
@pytest.mark.parametrize("m", [1, 13, 26, 40, 67])
@pytest.mark.parametrize("group_size", [-1, 32, 64, 128])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_marlin_gemm(gpu_device, k, n, m, group_size, dtype):
    pr = helpers.make_w4a16_problem(0, m, k, n, 4, group_size, False, dtype)
    c = hip_gemm(pr, m, n, k, 4, gpu_device)
    ref = oracle.gptq_marlin_gemm(pr["a"], pr["marlin_q_w"], pr["marlin_s"], None, None, 4, m, n, k)
    ref2 = pr["a"].float() @ pr["w_ref"].float()  # the reference test's own checker
    assert not torch.isnan(c.float()).any()
    assert ref_math.compute_max_diff(c, ref) < 0.04
    assert ref_math.compute_max_diff(c, ref2) < 0.04
    # fp32 group scaling is tighter than the reference's tolerance by an order of magnitude
    assert ref_math.compute_max_diff(c, ref) < 6e-3


TALL_CASES = [  # (rows per wave / 16, k groups per workgroup, M values that pick or fit the tile)
    (1, 1, [1, 9, 16]), (1, 2, [3, 16]), (1, 4, [1, 12]),
    (2, 1, [17, 33]), (2, 2, [20, 64]), (2, 4, [31, 47]),
    (4, 1, [40, 65]), (4, 2, [64, 130]), (4, 4, [33, 100]),
    (8, 1, [129, 300]),   # the 128-row prescale (prefill) tile
]


@pytest.mark.parametrize("mt,wk,ms", TALL_CASES)
@pytest.mark.parametrize("k,n", [(1024, 64), (2048, 448)])
@pytest.mark.parametrize("group_size", [-1, 128])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
@pytest.mark.parametrize("bits", [4, 8])
def test_marlin_gemm_tall_variants(gpu_device, monkeypatch, mt, wk, ms, k, n, group_size, dtype, bits):
    """every instantiation of the default (tall register tile) kernel, forced through its
    development knobs: tile height x in-workgroup k split, ragged M, with and without split-K;
    4-bit and 8-bit (W8A16) codes (the plan maps 8-bit requests for tiles it does not have --
    128 rows, 64 rows x 4 k groups -- onto the nearest one it has)"""
    monkeypatch.setenv("NMV_W4_TALL_MT", str(mt))
    monkeypatch.setenv("NMV_W4_TALL_WK", str(wk))
    for m in ms:
        for splits in (None, 2):
            if splits:
                monkeypatch.setenv("NMV_W4_SPLITS", str(splits))
            else:
                monkeypatch.delenv("NMV_W4_SPLITS", raising=False)
            pr = helpers.make_w4a16_problem(5, m, k, n, bits, group_size, False, dtype)
            c = hip_gemm(pr, m, n, k, bits, gpu_device)
            ref = oracle.gptq_marlin_gemm(pr["a"], pr["marlin_q_w"], pr["marlin_s"], None, None, bits, m, n, k)
            assert not torch.isnan(c.float()).any()
            assert ref_math.compute_max_diff(c, ref) < 6e-3, (m, splits)


@pytest.mark.parametrize("m", [1, 40])
def test_marlin_gemm_16row_kernel_still_covered(gpu_device, monkeypatch, m):
    """NMV_W4_TALL=0 routes group-128 problems to the 16-row kernel (the path of groups 32/64,
    act-order and K % 256 != 0): same answer within rounding"""
    k, n = 1024, 448
    pr = helpers.make_w4a16_problem(6, m, k, n, 4, 128, False, torch.bfloat16)
    c_tall = hip_gemm(pr, m, n, k, 4, gpu_device)
    monkeypatch.setenv("NMV_W4_TALL", "0")
    c_16 = hip_gemm(pr, m, n, k, 4, gpu_device)
    ref = oracle.gptq_marlin_gemm(pr["a"], pr["marlin_q_w"], pr["marlin_s"], None, None, 4, m, n, k)
    assert ref_math.compute_max_diff(c_16, ref) < 6e-3
    assert ref_math.compute_max_diff(c_tall, c_16) < 6e-3


@pytest.mark.parametrize("k,n", [(1024, 256), (640, 448)])
@pytest.mark.parametrize("m", [1, 26])
@pytest.mark.parametrize("group_size", [32, 128])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_marlin_gemm_act_order_k_full(gpu_device, k, n, m, group_size, dtype):
    pr = helpers.make_w4a16_problem(1, m, k, n, 4, group_size, True, dtype)
    c = hip_gemm(pr, m, n, k, 4, gpu_device, is_k_full=True)
    ref = oracle.gptq_marlin_gemm(pr["a"], pr["marlin_q_w"], pr["marlin_s"], pr["g_idx"],
                                  pr["sort_indices"], 4, m, n, k)
    assert ref_math.compute_max_diff(c, ref) < 0.04


@pytest.mark.parametrize("m", [1, 16, 64])
@pytest.mark.parametrize("k,n", [(4096, 6144), (4096, 4096), (14336, 4096)])
def test_marlin_gemm_llama_shapes(gpu_device, m, k, n):
    """the real decode shapes of Llama-3-8B (BASELINE.json configs[2]), group 128, bf16"""
    pr = helpers.make_w4a16_problem(2, m, k, n, 4, 128, False, torch.bfloat16)
    c = hip_gemm(pr, m, n, k, 4, gpu_device)
    ref = oracle.gptq_marlin_gemm(pr["a"], pr["marlin_q_w"], pr["marlin_s"], None, None, 4, m, n, k)
    assert ref_math.compute_max_diff(c, ref) < 6e-3


@pytest.mark.parametrize("tp", [2, 4, 8])
@pytest.mark.parametrize("m", [1, 64])
def test_marlin_gemm_tp_shard_shapes(gpu_device, tp, m):
    """the per-rank shapes of Llama-3-8B under tensor parallelism (column-parallel qkv / gate_up:
    N / tp; row-parallel o / down: K / tp) -- what `bench.py --gpus tp` launches on every rank"""
    for k, n in [(4096, 6144 // tp), (4096 // tp, 4096), (4096, 28672 // tp), (14336 // tp, 4096)]:
        pr = helpers.make_w4a16_problem(7, m, k, n, 4, 128, False, torch.bfloat16)
        c = hip_gemm(pr, m, n, k, 4, gpu_device)
        ref = oracle.gptq_marlin_gemm(pr["a"], pr["marlin_q_w"], pr["marlin_s"], None, None, 4, m, n, k)
        assert ref_math.compute_max_diff(c, ref) < 6e-3, (tp, k, n)


@pytest.mark.parametrize("m", [1, 64])
def test_marlin_gemm_llama70b_tp8_shapes(gpu_device, m):
    """BASELINE.json configs[4]: Llama-3-70B TP=8 per-rank projections"""
    for k, n in [(8192, 1280), (1024, 8192), (8192, 7168), (3584, 8192)]:
        pr = helpers.make_w4a16_problem(8, m, k, n, 4, 128, False, torch.bfloat16)
        c = hip_gemm(pr, m, n, k, 4, gpu_device)
        ref = oracle.gptq_marlin_gemm(pr["a"], pr["marlin_q_w"], pr["marlin_s"], None, None, 4, m, n, k)
        assert ref_math.compute_max_diff(c, ref) < 6e-3, (k, n)


@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
@pytest.mark.parametrize("group_size", [-1, 128])
def test_marlin_gemm_one_hot_is_exact_dequant(gpu_device, dtype, group_size):
    """size-independent property: a one-hot activation row selects one weight row, so the output
    must equal round((q - 8) * s) BIT FOR BIT -- checks every nibble position and scale index."""
    k, n, m = 1024, 1088, 64
    pr = helpers.make_w4a16_problem(3, m, k, n, 4, group_size, False, dtype)
    g = torch.Generator().manual_seed(3)
    rows = torch.randperm(k, generator=g)[:m]
    a = torch.zeros((m, k), dtype=dtype)
    a[torch.arange(m), rows] = 1.0
    pr["a"] = a
    c = hip_gemm(pr, m, n, k, 4, gpu_device)
    q = oracle.marlin_unpack(pr["marlin_q_w"], k, n, 4).float() - 8
    gs = k if group_size == -1 else group_size
    # natural-order scales: invert marlin_permute_scales through the oracle GEMM on identity rows
    ref = oracle.gptq_marlin_gemm(a, pr["marlin_q_w"], pr["marlin_s"], None, None, 4, m, n, k)
    assert torch.equal(c, ref)
    assert torch.equal(c.float() != 0, (q[rows] != 0))


def test_marlin_gemm_linearity(gpu_device):
    """full Llama gate_up size (K=4096, N=28672): C(a1 + a2) == C(a1) + C(a2) within rounding"""
    k, n, m = 4096, 28672, 8
    g = torch.Generator().manual_seed(4)
    q_w = torch.randint(0, 16, (k, n), generator=g, dtype=torch.int32)
    mq = ref_math.marlin_weights(q_w, k, n, 4)
    s = (torch.rand((k // 128, n), generator=g) * 0.01 + 0.001).to(torch.bfloat16)
    ms = ref_math.marlin_permute_scales(s, k, n, 128)
    a1 = torch.randn((m, k), generator=g).to(torch.bfloat16)
    a2 = torch.randn((m, k), generator=g).to(torch.bfloat16)
    asum = (a1.float() + a2.float()).to(torch.bfloat16)
    e = torch.empty(0, dtype=torch.int32)
    mk = lambda a: dict(a=a, marlin_q_w=mq, marlin_s=ms, g_idx=e, sort_indices=e)  # noqa: E731
    c1 = hip_gemm(mk(a1), m, n, k, 4, gpu_device).float()
    c2 = hip_gemm(mk(a2), m, n, k, 4, gpu_device).float()
    cs = hip_gemm(mk(asum), m, n, k, 4, gpu_device).float()
    # asum is rounded to bf16 (rel 2^-9 per element): compare in the mean
    assert ((cs - (c1 + c2)).abs().mean() / (c1 + c2).abs().mean()) < 2e-2
    # spot-check 64 random columns of the big problem against the oracle on a column slice
    cols = torch.randperm(n // 64, generator=g)[:2]
    for cb in cols.tolist():
        sl = slice(cb * 64, cb * 64 + 64)
        w = (q_w[:, sl].float() - 8) * s.float().repeat_interleave(128, dim=0)[:, sl]
        ref = a1.float() @ w.to(torch.bfloat16).float()
        assert ref_math.compute_max_diff(c1[:, sl], ref) < 6e-3


@pytest.mark.parametrize("m", [1, 64])
def test_marlin_gemm_full_gate_up(gpu_device, m):
    """the whole Llama-3-8B gate_up projection (K = 4096, N = 28672: 448 chunks, the widest launch of the decode step)
    at the bench's batch sizes, every column against the dequantise-then-matmul definition
    (quantize_weights' w_ref = (q - 8) * s rounded to the model dtype, test_marlin_gemm.py:172-179)"""
    k, n = 4096, 28672
    g = torch.Generator().manual_seed(40 + m)
    q_w = torch.randint(0, 16, (k, n), generator=g, dtype=torch.int32)
    mq = ref_math.marlin_weights(q_w, k, n, 4)
    s = (torch.rand((k // 128, n), generator=g) * 0.01 + 0.001).to(torch.bfloat16)
    ms = ref_math.marlin_permute_scales(s, k, n, 128)
    a = torch.randn((m, k), generator=g).to(torch.bfloat16)
    e = torch.empty(0, dtype=torch.int32)
    c = hip_gemm(dict(a=a, marlin_q_w=mq, marlin_s=ms, g_idx=e, sort_indices=e), m, n, k, 4, gpu_device).float()
    ref = torch.empty((m, n))
    for c0 in range(0, n, 4096):   # column slabs keep the dense fp32 weight at 64 MB
        sl = slice(c0, c0 + 4096)
        w = ((q_w[:, sl] - 8).float() * s.float().repeat_interleave(128, dim=0)[:, sl]).to(torch.bfloat16).float()
        ref[:, sl] = a.float() @ w
    assert ref_math.compute_max_diff(c, ref) < 6e-3
    # no chunk is off by itself (a wrong chunk would drown in the mean over 28672 columns)
    per_chunk = ((c - ref).abs().view(m, n // 64, 64).mean(dim=(0, 2)) /
                 ref.abs().view(m, n // 64, 64).mean(dim=(0, 2)).clamp_min(1e-6))
    assert per_chunk.max().item() < 3e-2, per_chunk.argmax().item()


def test_marlin_gemm_deterministic(gpu_device):
    pr = helpers.make_w4a16_problem(5, 16, 4096, 4096, 4, 128, False, torch.bfloat16)
    c1 = hip_gemm(pr, 16, 4096, 4096, 4, gpu_device)
    c2 = hip_gemm(pr, 16, 4096, 4096, 4, gpu_device)
    assert torch.equal(c1, c2)  # fixed-order split-K reduction


def test_marlin_gemm_arg_checks(gpu_device):
    from neural_magic_vllm_amd import _custom_ops as ops
    pr = helpers.make_w4a16_problem(0, 4, 128, 64, 4, -1, False, torch.half)
    d = gpu_device
    e = torch.empty(0, dtype=torch.int32, device=d)
    ws = torch.zeros(16, dtype=torch.int32, device=d)
    with pytest.raises(RuntimeError, match="size_m"):
        ops.gptq_marlin_gemm(pr["a"].to(d), pr["marlin_q_w"].to(d), pr["marlin_s"].to(d), e, e, ws, 4, 5, 64, 128, True)
    with pytest.raises(RuntimeError, match="num_bits"):
        ops.gptq_marlin_gemm(pr["a"].to(d), pr["marlin_q_w"].to(d), pr["marlin_s"].to(d), e, e, ws, 3, 4, 64, 128, True)
    with pytest.raises(RuntimeError, match="workspace"):
        ops.gptq_marlin_gemm(pr["a"].to(d), pr["marlin_q_w"].to(d), pr["marlin_s"].to(d), e, e, ws[:8], 4, 4, 64, 128, True)


@pytest.mark.parametrize("m", [1, 16, 33, 64, 130, 512, 1100])
@pytest.mark.parametrize("group_size", [128, -1])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_gemm_silu_mul_epilogue_matches_separate_ops(gpu_device, monkeypatch, m, group_size, dtype):
    """gate_up GEMM with silu_and_mul folded into the epilogue, on column-interleaved weights, against
    gptq_marlin_gemm on the original weight followed by silu_and_mul: bit for bit, on every row tile
    (16 / 32 / 64 / 128 rows per wave).  The fused form never splits K, so the comparison pins the
    plain GEMM to one slab as well (otherwise the fp32 summation order differs)."""
    monkeypatch.setenv("NMV_W4_SPLITS", "1")
    from neural_magic_vllm_amd import _custom_ops as ops
    d = gpu_device
    k, inter = 512, 8192
    n = 2 * inter
    g = torch.Generator().manual_seed(5)
    a = (torch.randn((m, k), generator=g) * 0.5).to(dtype)
    w = (torch.randn((k, n), generator=g) * 0.1).to(dtype)
    _, q_w, s, _, _ = ref_math.quantize_weights(w, 4, group_size, False, g)
    s = s.to(dtype)
    e = torch.empty(0, dtype=torch.int32, device=d)

    def marlin(q, sc):
        mq = ops.gptq_marlin_repack(ref_math.gptq_pack(q, 4, k, n).to(d), e, k, n, 4)
        ms = ref_math.marlin_permute_scales(sc, k, n, group_size).to(d)
        return mq, ms

    def interleave(t):
        return t.reshape(t.shape[0], 2, n // 64, 32).transpose(1, 2).reshape(t.shape[0], n).contiguous()

    ws = torch.zeros(n // 64 * 16, dtype=torch.int32, device=d)
    mq, ms = marlin(q_w, s)
    full = ops.gptq_marlin_gemm(a.to(d), mq, ms, e, e, ws, 4, m, n, k, True)
    ref = torch.empty((m, inter), dtype=dtype, device=d)
    ops.silu_and_mul(ref, full)
    mq_i, ms_i = marlin(interleave(q_w), interleave(s))
    got = ops.gptq_marlin_gemm_silu_mul(a.to(d), mq_i, ms_i, ws, m, n, k)
    assert got.shape == (m, inter)
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16))
    assert int(ws.abs().sum()) == 0


@pytest.mark.parametrize("m", [1, 5, 16, 64, 200])
@pytest.mark.parametrize("k,n", [(4096, 4096), (14336, 4096), (512, 512), (1024, 8192),
                                 (512, 4096), (1792, 4096)])   # the last two: o / down of Llama-3-8B at TP = 8
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_deferred_split_k_matches_gemm_then_norm(gpu_device, m, k, n, dtype):
    """gptq_marlin_gemm_partial (fp32 slabs, no ticket / last-arriver pass) + fused_add_rms_norm_partial
    (sums the slabs in split order) against gptq_marlin_gemm + fused_add_rms_norm: the normalised
    output and the updated residual, bit for bit"""
    from neural_magic_vllm_amd import _custom_ops as ops
    d = gpu_device
    pr = helpers.make_w4a16_problem(7, m, k, n, 4, 128, False, dtype)
    a, mq, ms = pr["a"].to(d), pr["marlin_q_w"].to(d), pr["marlin_s"].to(d)
    g = torch.Generator().manual_seed(8)
    res = torch.randn((m, n), generator=g).to(dtype).to(d)
    w = (1 + 0.1 * torch.randn((n, ), generator=g)).to(dtype).to(d)
    e = torch.empty(0, dtype=torch.int32, device=d)
    ws = torch.zeros(max(n // 64 * 16, 16), dtype=torch.int32, device=d)
    out = ops.gptq_marlin_gemm(a, mq, ms, e, e, ws, 4, m, n, k, True)
    ref_res = res.clone()
    ops.fused_add_rms_norm(out, ref_res, w, 1e-5)
    splits = ops.gptq_marlin_gemm_partial_splits(m, n, k)
    slab = ops.gptq_marlin_gemm_partial(a, mq, ms, m, n, k)
    assert slab.shape == (splits, m, n) and slab.dtype == torch.float32
    got_res = res.clone()
    got = ops.fused_add_rms_norm_partial(slab, got_res, w, 1e-5)
    assert torch.equal(got_res.view(torch.int16), ref_res.view(torch.int16))
    assert torch.equal(got.view(torch.int16), out.view(torch.int16))


@pytest.mark.parametrize("m", [1, 19, 64])
@pytest.mark.parametrize("kv_cache_dtype", ["auto", "fp8"])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
@pytest.mark.parametrize("heads,kv_heads", [(8, 2), (4, 1)])   # (4, 1): the TP = 8 shard of Llama-3-8B
def test_deferred_split_k_qkv_rope_cache(gpu_device, m, kv_cache_dtype, dtype, heads, kv_heads):
    """qkv projection as fp32 slabs + rotary_embedding_and_cache_partial against gptq_marlin_gemm +
    rotary_embedding_and_cache: rotated qkv row and both caches, bit for bit"""
    from neural_magic_vllm_amd import _custom_ops as ops
    d = gpu_device
    hs, k = 128, 1024
    n = (heads + 2 * kv_heads) * hs
    pr = helpers.make_w4a16_problem(9, m, k, n, 4, 128, False, dtype)
    a, mq, ms = pr["a"].to(d), pr["marlin_q_w"].to(d), pr["marlin_s"].to(d)
    g = torch.Generator().manual_seed(10)
    block_size, num_blocks, max_pos = 16, 9, 2048
    cos_sin = torch.randn((max_pos, hs), generator=g).to(dtype).to(d)
    positions = torch.randint(0, max_pos, (m, ), generator=g).to(d)
    slots = torch.randperm(num_blocks * block_size, generator=g)[:m].to(d)
    if m > 3:
        slots[3] = -1
    cdt = torch.uint8 if kv_cache_dtype == "fp8" else dtype
    x = 16 // torch.tensor([], dtype=cdt).element_size()
    kv_scale = 0.5 if kv_cache_dtype == "fp8" else 1.0

    def caches():
        gen = torch.Generator().manual_seed(1)
        kc = torch.randint(0, 100, (num_blocks, kv_heads, hs // x, block_size, x), generator=gen)
        vc = torch.randint(0, 100, (num_blocks, kv_heads, hs, block_size), generator=gen)
        return kc.to(cdt).to(d), vc.to(cdt).to(d)

    e = torch.empty(0, dtype=torch.int32, device=d)
    ws = torch.zeros(n // 64 * 16, dtype=torch.int32, device=d)
    ref = ops.gptq_marlin_gemm(a, mq, ms, e, e, ws, 4, m, n, k, True)
    q, kk, v = ref.split([heads * hs, kv_heads * hs, kv_heads * hs], dim=-1)
    ref_kc, ref_vc = caches()
    ops.rotary_embedding_and_cache(positions, q, kk, v, hs, cos_sin, True, ref_kc, ref_vc, slots,
                                   kv_cache_dtype, kv_scale)
    slab = ops.gptq_marlin_gemm_partial(a, mq, ms, m, n, k)
    kc, vc = caches()
    got = ops.rotary_embedding_and_cache_partial(positions, slab, heads, kv_heads, hs, cos_sin, kc, vc, slots,
                                                 kv_cache_dtype, kv_scale, dtype)
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16))
    assert torch.equal(kc.view(torch.uint8), ref_kc.view(torch.uint8))
    assert torch.equal(vc.view(torch.uint8), ref_vc.view(torch.uint8))
    # without caches: rope only
    got2 = ops.rotary_embedding_and_cache_partial(positions, slab, heads, kv_heads, hs, cos_sin, None, None,
                                                  None, kv_cache_dtype, kv_scale, dtype)
    assert torch.equal(got2.view(torch.int16), ref.view(torch.int16))
