"""GPU: the MFMA-native W4 tensor (nmv_w4_native_repack) and its kernel (nmv_w4_native_gemm), which
GPTQMarlinLinearMethod uses for decode-sized calls.  Not ops of the reference: the layout is specified by
oracle/ref_math.w4_native_weights, the arithmetic is the Marlin kernels' (fp32 group scaling), so the checks
are: repack bit-equal to the specification (with and without the act-order row gather), GEMM against the
oracle's a @ w_ref at the reference's tolerance (test_marlin_gemm.py:172-179, < 0.04) and this repo's own
(< 6e-3), a one-hot row reproducing the rounded dequantised weight bit for bit, and the three modes bit-identical
to each other's op sequences (deferred slabs summed in split order; silu epilogue vs GEMM + silu_and_mul)."""
import pytest
import torch

from oracle import ref_math

pytestmark = pytest.mark.gpu


def rel_err(out, ref):
    return ((out.float() - ref.float()).abs().mean() / ref.float().abs().mean()).item()


def problem(seed, m, k, n, group_size, dtype):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn((m, k), generator=g).to(dtype)
    w = torch.randn((k, n), generator=g).to(dtype)
    w_ref, q_w, s, _, _ = ref_math.quantize_weights(w, 4, group_size, False)
    return a, q_w, s.to(dtype), w_ref.to(dtype)


@pytest.mark.parametrize("k,n", [(256, 64), (1024, 448), (512, 1088)])
@pytest.mark.parametrize("act_order", [False, True])
def test_native_repack_matches_specification(gpu_device, k, n, act_order):
    from neural_magic_vllm_amd import _custom_ops as ops
    g = torch.Generator().manual_seed(0)
    q_w = torch.randint(0, 16, (k, n), generator=g, dtype=torch.int32)
    packed = ref_math.gptq_pack(q_w, 4, k, n).to(gpu_device)
    perm = torch.randperm(k, generator=g).to(torch.int32) if act_order else None
    out = ops.w4_native_repack(packed, perm.to(gpu_device) if act_order else None, k, n).cpu()
    want = ref_math.w4_native_weights(q_w[perm.long()] if act_order else q_w)
    assert torch.equal(out, want)


def native_gemm(a, q_w, s, k, n, dev, mode=0):
    from neural_magic_vllm_amd import _custom_ops as ops
    packed = ref_math.gptq_pack(q_w, 4, k, n).to(dev)
    b = ops.w4_native_repack(packed, None, k, n)
    ws = torch.zeros(max(n // 64 * 16, 16), dtype=torch.int32, device=dev)
    out = ops.w4_native_gemm(a.to(dev), b, s.to(dev), ws, a.shape[0], n, k, mode)
    assert int(ws.abs().sum()) == 0, "the ticket array must be returned zeroed"
    return out


@pytest.mark.parametrize("k,n", [(256, 64), (1024, 448), (2048, 1088), (4096, 6144), (14336, 4096)])
@pytest.mark.parametrize("m", [1, 13, 16, 17, 32, 40, 64])
@pytest.mark.parametrize("group_size", [-1, 128])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_native_gemm(gpu_device, k, n, m, group_size, dtype):
    a, q_w, s, w_ref = problem(1, m, k, n, group_size, dtype)
    out = native_gemm(a, q_w, s, k, n, gpu_device).cpu()
    ref = (a.double() @ w_ref.double()).float()
    e = rel_err(out, ref)
    assert e < 0.04 and e < 6e-3, e


@pytest.mark.parametrize("mt,wk", [(1, 4), (1, 2), (1, 1), (2, 4), (2, 2), (2, 1), (4, 4), (4, 2), (4, 1)])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_native_gemm_every_tile_variant(gpu_device, monkeypatch, mt, wk, dtype):
    monkeypatch.setenv("NMV_W4_TALL_MT", str(mt))
    monkeypatch.setenv("NMV_W4_TALL_WK", str(wk))
    k, n, m = 2048, 448, 16 * mt - 3
    a, q_w, s, w_ref = problem(2, m, k, n, 128, dtype)
    out = native_gemm(a, q_w, s, k, n, gpu_device).cpu()
    assert rel_err(out, (a.double() @ w_ref.double()).float()) < 6e-3


@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
@pytest.mark.parametrize("group_size", [-1, 128])
def test_native_gemm_one_hot_is_exact_dequant(gpu_device, dtype, group_size):
    """a one-hot activation row selects one weight row: the output must be round((q - 8) * s) exactly"""
    k, n = 512, 448
    _, q_w, s, w_ref = problem(3, 1, k, n, group_size, dtype)
    rows = [0, 1, 7, 8, 31, 32, 127, 128, 255, 511]
    a = torch.zeros((len(rows), k), dtype=dtype)
    for i, r in enumerate(rows):
        a[i, r] = 1.0
    out = native_gemm(a, q_w, s, k, n, gpu_device).cpu()
    gs = k if group_size == -1 else group_size
    want = ((q_w[rows].float() - 8) * s.float()[[r // gs for r in rows]]).to(dtype)
    assert torch.equal(out.view(torch.int16), want.view(torch.int16))


@pytest.mark.parametrize("m", [1, 16, 33, 64])
@pytest.mark.parametrize("k,n", [(4096, 4096), (14336, 4096), (1024, 6144)])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_native_deferred_slabs_sum_to_the_gemm(gpu_device, m, k, n, dtype):
    """mode 2 (deferred reduction): the slabs summed in split order from +0 and rounded are the bits of mode 0"""
    a, q_w, s, _ = problem(4, m, k, n, 128, dtype)
    full = native_gemm(a, q_w, s, k, n, gpu_device, 0)
    slab = native_gemm(a, q_w, s, k, n, gpu_device, 2)
    acc = torch.zeros_like(slab[0])
    for i in range(slab.shape[0]):
        acc = acc + slab[i]
    assert torch.equal(acc.to(dtype).view(torch.int16), full.view(torch.int16))


@pytest.mark.parametrize("m", [1, 16, 64])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_native_silu_mul_epilogue_matches_separate_ops(gpu_device, m, dtype):
    from neural_magic_vllm_amd import _custom_ops as ops
    from neural_magic_vllm_amd.model_executor.layers.quantization.gptq_marlin import GPTQMarlinLinearMethod as LM
    # a projection wide enough that neither form splits K across workgroups (as the Marlin twin of this test,
    # test_gpu_w4a16.py: the fused form never splits, and a split plain GEMM sums its fp32 slabs in another order)
    k, inter = 512, 8192
    n = 2 * inter
    a, q_w, s, _ = problem(5, m, k, n, 128, dtype)
    plain = native_gemm(a, q_w, s, k, n, gpu_device, 0)
    want = torch.empty((m, inter), dtype=dtype, device=gpu_device)
    ops.silu_and_mul(want, plain)
    qi, si = LM._interleave_gate_up(q_w), LM._interleave_gate_up(s)
    got = native_gemm(a, qi, si, k, n, gpu_device, 1)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))


@pytest.mark.parametrize("m", [1, 16, 64])
def test_native_silu_mul_on_a_split_shape(gpu_device, m):
    """K = 4096, N = 7168 (112 chunks: Llama-3-8B gate_up at TP = 4): the plain GEMM splits K across workgroups and sums
    its fp32 slabs, the fused launch never splits, so the two agree up to fp32 summation order -- stated tolerance: the
    reference's (test_marlin_gemm.py:172-179, < 0.04 relative) and at most one ulp of the model dtype per element"""
    from neural_magic_vllm_amd import _custom_ops as ops
    from neural_magic_vllm_amd.model_executor.layers.quantization.gptq_marlin import GPTQMarlinLinearMethod as LM
    dtype = torch.bfloat16
    k, n = 4096, 7168
    a, q_w, s, _ = problem(9, m, k, n, 128, dtype)
    plain = native_gemm(a, q_w, s, k, n, gpu_device, 0)
    want = torch.empty((m, n // 2), dtype=dtype, device=gpu_device)
    ops.silu_and_mul(want, plain)
    got = native_gemm(a, LM._interleave_gate_up(q_w), LM._interleave_gate_up(s), k, n, gpu_device, 1)
    assert rel_err(got, want) < 1e-3
    # gate and up are each rounded to the model dtype before silu * up: an ulp (2^-8 relative) in either moves the product
    # by a few ulps of its own magnitude; products near zero are bounded absolutely (an ulp of gate there is an ulp of
    # the typical gate, not of the product)
    g, w = got.float(), want.float()
    bound = 4 * 2.0 ** -8 * w.abs() + 2.0 ** -8 * w.abs().mean()
    assert bool(((g - w).abs() <= bound).all())
    assert float((got.view(torch.int16) != want.view(torch.int16)).float().mean()) < 0.05


@pytest.mark.parametrize("m", [1, 16, 48, 64])
def test_native_gemm_full_gate_up(gpu_device, m):
    """the whole Llama-3-8B gate_up projection (K = 4096, N = 28672: 448 chunks) on the native tensor -- the widest
    launch of the decode step in the form it actually takes (resident at M <= 16, the hand-scheduled 64-row stage at
    M = 33 .. 64) -- every column against the dequantise-then-matmul definition, chunk by chunk (the Marlin twin:
    test_gpu_w4a16.py::test_marlin_gemm_full_gate_up), and twice for the fixed-order reductions"""
    from neural_magic_vllm_amd import _custom_ops as ops
    k, n = 4096, 28672
    g = torch.Generator().manual_seed(60 + m)
    q_w = torch.randint(0, 16, (k, n), generator=g, dtype=torch.int32)
    s = (torch.rand((k // 128, n), generator=g) * 0.01 + 0.001).to(torch.bfloat16)
    a = torch.randn((m, k), generator=g).to(torch.bfloat16)
    d = gpu_device
    b = ops.w4_native_repack(ref_math.gptq_pack(q_w, 4, k, n).to(d), None, k, n)
    ws = torch.zeros(n // 64 * 16, dtype=torch.int32, device=d)
    c1 = ops.w4_native_gemm(a.to(d), b, s.to(d), ws, m, n, k, 0)
    c2 = ops.w4_native_gemm(a.to(d), b, s.to(d), ws, m, n, k, 0)
    assert torch.equal(c1, c2)
    c = c1.float().cpu()
    ref = torch.empty((m, n))
    for c0 in range(0, n, 4096):   # column slabs keep the dense fp32 weight at 64 MB
        sl = slice(c0, c0 + 4096)
        w = ((q_w[:, sl] - 8).float() * s.float().repeat_interleave(128, dim=0)[:, sl]).to(torch.bfloat16).float()
        ref[:, sl] = a.float() @ w
    assert rel_err(c, ref) < 6e-3
    per_chunk = ((c - ref).abs().view(m, n // 64, 64).mean(dim=(0, 2)) /
                 ref.abs().view(m, n // 64, 64).mean(dim=(0, 2)).clamp_min(1e-6))
    assert per_chunk.max().item() < 3e-2, per_chunk.argmax().item()


@pytest.mark.parametrize("m", [5, 40, 64])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_wide_launch_with_an_odd_chunk_count(gpu_device, m, dtype):
    """a wide projection (>= 224 chunks: one k range per workgroup, the streamed 64-row stage from M = 33) whose chunk
    count is odd -- the last workgroup owns one real chunk and one past N -- on both tensors, against a @ w_ref"""
    from neural_magic_vllm_amd import _custom_ops as ops
    k, n = 512, 225 * 64
    a, q_w, s, w_ref = problem(7, m, k, n, 128, dtype)
    ref = (a.double() @ w_ref.double()).float()
    out = native_gemm(a, q_w, s, k, n, gpu_device).cpu()
    assert rel_err(out, ref) < 6e-3
    d = gpu_device
    e = torch.empty(0, dtype=torch.int32, device=d)
    mq = ops.gptq_marlin_repack(ref_math.gptq_pack(q_w, 4, k, n).to(d), e, k, n, 4)
    ms = ref_math.marlin_permute_scales(s, k, n, 128).to(d)
    ws = torch.zeros(n // 64 * 16, dtype=torch.int32, device=d)
    out2 = ops.gptq_marlin_gemm(a.to(d), mq, ms, e, e, ws, 4, m, n, k, True).cpu()
    assert rel_err(out2, ref) < 6e-3
    assert int(ws.abs().sum()) == 0


@pytest.mark.parametrize("k,n", [(8192, 1280), (1024, 8192), (8192, 7168), (3584, 8192),      # Llama-3-70B at TP = 8, per rank
                                 (4096, 3072), (2048, 4096), (4096, 14336), (7168, 4096)])    # Llama-3-8B at TP = 2
@pytest.mark.parametrize("m", [1, 64])
def test_native_gemm_tensor_parallel_shard_shapes(gpu_device, k, n, m):
    """the per-rank projections of BASELINE.json configs[4] (Llama-3-70B w4a16, TP = 8) and of Llama-3-8B at TP = 2 on
    the native tensor -- what every rank of a tensor-parallel decode step launches since the native copy is the default:
    the plain form against a @ w_ref, and the deferred slabs summing to it bit for bit"""
    a, q_w, s, w_ref = problem(9, m, k, n, 128, torch.bfloat16)
    out = native_gemm(a, q_w, s, k, n, gpu_device)
    assert rel_err(out.cpu(), (a.double() @ w_ref.double()).float()) < 6e-3
    slabs = native_gemm(a, q_w, s, k, n, gpu_device, 2)
    acc = torch.zeros_like(slabs[0])
    for i in range(slabs.shape[0]):
        acc = acc + slabs[i]
    assert torch.equal(acc.to(torch.bfloat16).view(torch.int16), out.view(torch.int16))
