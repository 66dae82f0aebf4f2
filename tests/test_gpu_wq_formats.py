"""GPU parity for the remaining weight-only formats (SURVEY.md section 8 rows A7-8bit/act-order,
A8, A9, A10, A14): GPTQ/exllama, AWQ, legacy Marlin, 8-bit Marlin, act-order on a K shard,
fp8-Marlin.  The reference has NO kernel-level tests for awq_gemm / gptq_gemm / marlin_gemm
(parity unpinned there); the oracle is the math definition the reference's
reconstruct kernels implement: w = round_dtype((q - z) * s), c = round_dtype(a @ w)."""
import pytest
import torch

import helpers
import oracle
from oracle import ref_math

pytestmark = pytest.mark.gpu


def rel_err(out, ref):
    return ref_math.compute_max_diff(out, ref).item()


@pytest.mark.parametrize("bits", [4, 8, 2, 3])
@pytest.mark.parametrize("group_size", [128, 32, -1])
@pytest.mark.parametrize("m", [1, 13, 67])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
@pytest.mark.parametrize("mode", ["plain", "g_idx", "exllama_act_order"])
def test_gptq_gemm(gpu_device, bits, group_size, m, dtype, mode):
    from neural_magic_vllm_amd import _custom_ops as ops
    k, n = 512, 448
    g = torch.Generator().manual_seed(0)
    w = torch.randn((k, n), generator=g).to(dtype)
    a = torch.randn((m, k), generator=g).to(dtype)
    q, z, s = ref_math.quantize_asym(w, bits, group_size)
    gs = k if group_size == -1 else group_size
    g_idx = (torch.arange(k) // gs).to(torch.int32)
    if mode != "plain" and group_size != -1:
        rp = torch.randperm(k, generator=g)
        q, g_idx = q[rp], g_idx[rp]  # act-order: rows shuffled, g_idx says which group each row has
    w_ref = ref_math.gptq_reference_weight(q, z, s, g_idx, dtype)
    ref = (a.float() @ w_ref.float()).to(dtype)
    d = gpu_device
    if bits == 3:   # 32 codes per three words, along K for the weights and along N for the zero points
        qweight = ref_math.pack_3bit_stream(q, 0).to(d)
        qzeros = ref_math.pack_3bit_stream(z - 1, 1).to(d)
    else:
        qweight = ref_math.gptq_pack(q, bits, k, n).to(d)
        qzeros = ref_math.pack_cols(z - 1, bits).to(d)  # stored as zero - 1
    scales = s.to(d)
    if mode == "exllama_act_order":
        # GPTQLinearMethod.apply (gptq.py:211-225): g_idx <- argsort(g_idx); gptq_shuffle; gemm
        perm = torch.argsort(g_idx).to(torch.int32).to(d)
        ops.gptq_shuffle(qweight, perm, bits)
        out = ops.gptq_gemm(a.to(d), qweight, qzeros, scales, perm, True, bits)
    elif mode == "plain":
        e = torch.empty(0, dtype=torch.int32, device=d)
        ops.gptq_shuffle(qweight, e, bits)
        out = ops.gptq_gemm(a.to(d), qweight, qzeros, scales, e, True, bits)
    else:
        out = ops.gptq_gemm(a.to(d), qweight, qzeros, scales, g_idx.to(d), False, bits)
    assert rel_err(out.cpu(), ref) < 5e-3


@pytest.mark.parametrize("group_size", [128, 64, 32])
@pytest.mark.parametrize("m", [1, 16, 70])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_awq_gemm_and_dequantize(gpu_device, group_size, m, dtype):
    from neural_magic_vllm_amd import _custom_ops as ops
    k, n = 1024, 512
    g = torch.Generator().manual_seed(1)
    w = torch.randn((k, n), generator=g).to(dtype)
    a = torch.randn((m, k), generator=g).to(dtype)
    q, z, s = ref_math.quantize_asym(w, 4, group_size)
    w_ref = ref_math.awq_reference_weight(q, z, s, group_size, dtype)
    d = gpu_device
    qweight = ref_math.pack_cols(q, 4, ref_math.AWQ_NIBBLE_OF_COLUMN).to(d)  # [K, N/8]
    qzeros = ref_math.pack_cols(z, 4, ref_math.AWQ_NIBBLE_OF_COLUMN).to(d)  # [G, N/8]
    deq = ops.awq_dequantize(qweight, s.to(d), qzeros, 0, 0, 0)
    assert deq.shape == (k, n)
    assert torch.equal(deq.cpu(), w_ref)  # (q - z) * s with one rounding: bit exact
    # AWQLinearMethod.apply passes (x, qweight, scales, qzeros, pack_factor) (awq.py:172-173)
    out = ops.awq_gemm(a.to(d), qweight, s.to(d), qzeros, 8)
    assert rel_err(out.cpu(), (a.float() @ w_ref.float()).to(dtype)) < 5e-3


@pytest.mark.parametrize("group_size", [-1, 128])
@pytest.mark.parametrize("m", [1, 33])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_legacy_marlin_gemm(gpu_device, group_size, m, dtype):
    from neural_magic_vllm_amd import _custom_ops as ops
    k, n = 1024, 512
    pr = helpers.make_w4a16_problem(0, m, k, n, 4, group_size, False, dtype)
    d = gpu_device
    ws = torch.zeros(n // 64 * 16, dtype=torch.int32, device=d)
    out = ops.marlin_gemm(pr["a"].to(d), pr["marlin_q_w"].to(d), pr["marlin_s"].to(d), ws, m, n, k)
    ref = oracle.gptq_marlin_gemm(pr["a"], pr["marlin_q_w"], pr["marlin_s"], None, None, 4, m, n, k)
    assert rel_err(out.cpu(), ref) < 6e-3


@pytest.mark.parametrize("k,n", [(128, 64), (1024, 448), (640, 1088)])
@pytest.mark.parametrize("m", [1, 26, 67])
@pytest.mark.parametrize("group_size", [-1, 32, 128])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_marlin_gemm_8bit(gpu_device, k, n, m, group_size, dtype):
    """num_bits = 8 (test_marlin_gemm.py:126-179 with MARLIN_SUPPORTED_NUM_BITS = [4, 8])"""
    from neural_magic_vllm_amd import _custom_ops as ops
    pr = helpers.make_w4a16_problem(0, m, k, n, 8, group_size, False, dtype)
    d = gpu_device
    e = torch.empty(0, dtype=torch.int32, device=d)
    ws = torch.zeros(n // 64 * 16, dtype=torch.int32, device=d)
    out = ops.gptq_marlin_gemm(pr["a"].to(d), pr["marlin_q_w"].to(d), pr["marlin_s"].to(d), e, e, ws,
                               8, m, n, k, True)
    ref = oracle.gptq_marlin_gemm(pr["a"], pr["marlin_q_w"], pr["marlin_s"], None, None, 8, m, n, k)
    assert rel_err(out.cpu(), ref) < 0.04
    assert rel_err(out.cpu(), pr["a"].float() @ pr["w_ref"].float()) < 0.04
    assert rel_err(out.cpu(), ref) < 5e-3


@pytest.mark.parametrize("bits", [4, 8])
@pytest.mark.parametrize("m", [1, 26])
@pytest.mark.parametrize("group_size", [32, 128])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_marlin_gemm_act_order_not_k_full(gpu_device, bits, m, group_size, dtype):
    """is_k_full = False: the general act-order path (test_marlin_gemm.py K_FULL_OPTS)"""
    from neural_magic_vllm_amd import _custom_ops as ops
    k, n = 1024, 256
    pr = helpers.make_w4a16_problem(1, m, k, n, bits, group_size, True, dtype)
    d = gpu_device
    ws = torch.zeros(n // 64 * 16, dtype=torch.int32, device=d)
    out = ops.gptq_marlin_gemm(pr["a"].to(d), pr["marlin_q_w"].to(d), pr["marlin_s"].to(d),
                               pr["g_idx"].to(d), pr["sort_indices"].to(d), ws, bits, m, n, k, False)
    ref = oracle.gptq_marlin_gemm(pr["a"], pr["marlin_q_w"], pr["marlin_s"], pr["g_idx"],
                                  pr["sort_indices"], bits, m, n, k)
    assert rel_err(out.cpu(), ref) < 5e-3
    assert rel_err(out.cpu(), pr["a"].float() @ pr["w_ref"].float()) < 0.04


@pytest.mark.parametrize("m", [1, 26, 67])
@pytest.mark.parametrize("k,n", [(128, 64), (1024, 448)])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_fp8_marlin_gemm(gpu_device, m, k, n, dtype):
    """tests/kernels/test_marlin_gemm.py:238-304: fp8 weights packed 4 per int32
    (pack_fp8_to_int32, marlin_utils.py:227-247), repacked with bits = 8, channelwise scales."""
    from neural_magic_vllm_amd import _custom_ops as ops
    g = torch.Generator().manual_seed(0)
    a = torch.randn((m, k), generator=g).to(dtype)
    w = torch.randn((k, n), generator=g).to(dtype)
    scale = (w.float().abs().max() / 448.0)
    wq = (w.float() / scale).clamp(-448, 448).to(torch.float8_e4m3fn)  # [K, N]
    w_ref = (wq.float() * scale).to(dtype)
    # pack_fp8_to_int32: 4 consecutive K bytes per int32 -> GPTQ-style [K/4, N]
    by = wq.view(torch.uint8).to(torch.int32).reshape(k // 4, 4, n)
    packed = by[:, 0] | (by[:, 1] << 8) | (by[:, 2] << 16) | (by[:, 3] << 24)
    d = gpu_device
    e = torch.empty(0, dtype=torch.int32, device=d)
    mq = ops.gptq_marlin_repack(packed.contiguous().to(d), e, k, n, 8)
    scales = ref_math.marlin_permute_scales(scale.to(dtype).repeat(1, n), k, n, -1).to(d)
    ws = torch.zeros(n // 64 * 16, dtype=torch.int32, device=d)
    out = ops.fp8_marlin_gemm(a.to(d), mq, scales, ws, 8, m, n, k)
    assert rel_err(out.cpu(), a.float() @ w_ref.float()) < 0.04
    assert rel_err(out.cpu(), (a.float() @ w_ref.float()).to(dtype)) < 5e-3


@pytest.mark.parametrize("case", [0, 1, 2])
def test_fp8_marlin_gemm_reference_recipe_fixture(gpu_device, case):
    """the inputs the reference's own test_fp8_marlin_gemm drew and the output it expected
    (tools/make_golden_fp8_marlin.py -> tests/golden/fp8_marlin_*.npz), judged by the reference's checker
    (compute_max_diff < 0.04, test_marlin_gemm.py:300-304) and held to the pinned oracle"""
    import os

    import numpy as np

    import helpers
    import oracle
    from neural_magic_vllm_amd import _custom_ops as ops
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"fp8_marlin_{case}.npz"))
    dt = getattr(torch, str(g["dtype"]))
    m, n, k = int(g["m"]), int(g["n"]), int(g["k"])
    a = helpers.from_np(g["a"], dt)
    mq, ms = torch.from_numpy(g["marlin_q"].copy()), helpers.from_np(g["marlin_s"], dt)
    d = gpu_device
    ws = torch.zeros(max(n // 64 * 16, 16), dtype=torch.int32, device=d)
    out = ops.fp8_marlin_gemm(a.to(d), mq.to(d), ms.to(d), ws, 8, m, n, k).cpu()
    ref = helpers.from_np(g["output_ref"], dt)
    assert ((out.float() - ref.float()).abs().mean() / ref.float().abs().mean()).item() < 0.04
    assert rel_err(out, oracle.fp8_marlin_gemm(a, mq, ms, m, n, k)) < 5e-3


@pytest.mark.parametrize("k,n", [(256, 64), (1024, 448), (2048, 128)])
def test_awq_marlin_repack_is_the_marlin_layout(gpu_device, k, n):
    """AWQ words -> Marlin tensor: bit-equal to marlin_weights() of the unpacked codes"""
    from neural_magic_vllm_amd import _custom_ops as ops
    g = torch.Generator().manual_seed(2)
    q = torch.randint(0, 16, (k, n), generator=g, dtype=torch.int32)
    qweight = ref_math.pack_cols(q, 4, ref_math.AWQ_NIBBLE_OF_COLUMN).to(gpu_device)
    out = ops.awq_marlin_repack(qweight, k, n).cpu()
    assert torch.equal(out, ref_math.marlin_weights(q, k, n, 4))


@pytest.mark.parametrize("m", [1, 16, 40, 70, 300])
@pytest.mark.parametrize("k,n", [(1024, 448), (2048, 64)])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_marlin_zp_gemm_matches_awq(gpu_device, m, k, n, dtype):
    """asymmetric (zero-point) weights on the Marlin kernel: same answer as the AWQ definition
    w = (q - z) * s, for every tile height the plan picks (16 / 32 / 64 / 128 rows)"""
    from neural_magic_vllm_amd import _custom_ops as ops
    g = torch.Generator().manual_seed(5)
    w = torch.randn((k, n), generator=g).to(dtype)
    a = torch.randn((m, k), generator=g).to(dtype)
    q, z, s = ref_math.quantize_asym(w, 4, 128)
    w_ref = ref_math.awq_reference_weight(q, z, s, 128, dtype)
    d = gpu_device
    mq = ref_math.marlin_weights(q, k, n, 4).to(d)
    ms = ref_math.marlin_permute_scales(s, k, n, 128).to(d)
    mz = ref_math.marlin_permute_scales(z.to(dtype), k, n, 128).to(d)
    ws = torch.zeros(n // 64 * 16, dtype=torch.int32, device=d)
    out = ops.marlin_zp_gemm(a.to(d), mq, ms, mz, ws, m, n, k)
    assert int(ws.abs().sum()) == 0
    assert rel_err(out.cpu(), (a.float() @ w_ref.float()).to(dtype)) < 5e-3
