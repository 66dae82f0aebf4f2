"""GPU: the prompt-sized W4A16 kernel on the native tensor (csrc/w4a16_prefill.hip) behind nmv_w4_native_gemm for calls
of 65 rows and more, group 128.  Its arithmetic is the reference's: w = round((q - 8) * s) in the model dtype, fp32
accumulation (gptq_marlin.cu:267-278).  Checks: the oracle's a @ w_ref at the reference's tolerance
(test_marlin_gemm.py:172-179, < 0.04) and this repo's own (< 6e-3), a one-hot row exact, ragged tiles in M (M % 256),
N (N % 128 == 64) and few stages, every split-K count bit-consistent with its own slabs, the silu epilogue against the
two ops it replaces, run-to-run bit-identity."""
import pytest
import torch

from oracle import ref_math

pytestmark = pytest.mark.gpu


def rel_err(out, ref):
    return ((out.float() - ref.float()).abs().mean() / ref.float().abs().mean()).item()


def problem(seed, m, k, n, dtype):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn((m, k), generator=g).to(dtype)
    w = torch.randn((k, n), generator=g).to(dtype)
    w_ref, q_w, s, _, _ = ref_math.quantize_weights(w, 4, 128, False)
    return a, q_w, s.to(dtype), w_ref.to(dtype)


def native_gemm(a, q_w, s, k, n, dev, mode=0):
    from neural_magic_vllm_amd import _custom_ops as ops
    b = ops.w4_native_repack(ref_math.gptq_pack(q_w, 4, k, n).to(dev), None, k, n)
    ws = torch.zeros(max(n // 64 * 16, 64), dtype=torch.int32, device=dev)
    out = ops.w4_native_gemm(a.to(dev), b, s.to(dev), ws, a.shape[0], n, k, mode)
    assert int(ws.abs().sum()) == 0, "the ticket array must be returned zeroed"
    return out


@pytest.fixture(params=["tile128", "tile256"])
def prefill(monkeypatch, request):
    """every call of 65 rows and more takes the prefill kernel, with its 128 x 128 or its 256 x 256 tile"""
    monkeypatch.setenv("NMV_W4P", "2")
    monkeypatch.setenv("NMV_W4P_MIN_M", "65")
    monkeypatch.setenv("NMV_W4P_TILE", "1" if request.param == "tile128" else "2")
    yield monkeypatch


def test_prefill_plan_is_taken(gpu_device, prefill):
    from neural_magic_vllm_amd import _lib
    lib = _lib.load()
    prefill.setenv("NMV_W4P_SPLITS", "4")
    assert lib.nmv_w4_native_gemm_splits(512, 4096, 4096, 32) == 4
    prefill.setenv("NMV_W4P_SPLITS", "2")
    assert lib.nmv_w4_native_gemm_splits(300, 4096, 4096, 32) == 2


@pytest.mark.parametrize("k,n", [(256, 64), (384, 128), (1024, 448), (2048, 1088), (4096, 6144), (14336, 4096)])
@pytest.mark.parametrize("m", [65, 128, 255, 256, 257, 512, 700])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_prefill_gemm(gpu_device, prefill, k, n, m, dtype):
    a, q_w, s, w_ref = problem(21, m, k, n, dtype)
    out = native_gemm(a, q_w, s, k, n, gpu_device).cpu()
    ref = (a.double() @ w_ref.double()).float()
    e = rel_err(out, ref)
    assert e < 0.04 and e < 6e-3, e
    assert (out.float() - ref).abs().max().item() < 0.04 * ref.abs().max().item() + 1e-2


@pytest.mark.parametrize("splits", [1, 2, 4, 8])
@pytest.mark.parametrize("m", [100, 512])
def test_prefill_gemm_every_split_count(gpu_device, prefill, splits, m):
    prefill.setenv("NMV_W4P_SPLITS", str(splits))
    prefill.setenv("NMV_W4P_MAX_SPLITS", "64")
    k, n = 4096, 576
    a, q_w, s, w_ref = problem(22, m, k, n, torch.bfloat16)
    out = native_gemm(a, q_w, s, k, n, gpu_device)
    assert rel_err(out.cpu(), (a.double() @ w_ref.double()).float()) < 6e-3
    slab = native_gemm(a, q_w, s, k, n, gpu_device, 2)
    assert slab.shape[0] == splits
    acc = torch.zeros_like(slab[0])
    for i in range(splits):
        acc = acc + slab[i]
    assert torch.equal(acc.to(torch.bfloat16).view(torch.int16), out.view(torch.int16))
    again = native_gemm(a, q_w, s, k, n, gpu_device)
    assert torch.equal(again.view(torch.int16), out.view(torch.int16))


@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_prefill_gemm_one_hot_is_exact_dequant(gpu_device, prefill, dtype):
    """a one-hot activation row selects one weight row: the output must be round((q - 8) * s) exactly"""
    k, n, m = 512, 448, 300
    _, q_w, s, _ = problem(23, 1, k, n, dtype)
    rows = [(37 * i + i // 64) % k for i in range(m)]
    a = torch.zeros((m, k), dtype=dtype)
    for i, r in enumerate(rows):
        a[i, r] = 1.0
    out = native_gemm(a, q_w, s, k, n, gpu_device).cpu()
    want = ((q_w[rows].float() - 8) * s.float()[[r // 128 for r in rows]]).to(dtype)
    assert torch.equal(out.view(torch.int16), want.view(torch.int16))


def test_prefill_gemm_equals_the_dequantised_matmul_in_fp32_order_bound(gpu_device, prefill):
    """the weights the kernel multiplies ARE the reference's w_ref (the model-dtype rounding of (q - 8) * s): with
    activations that are small integers every product and partial sum is exact in fp32, so the result must equal
    a @ w_ref computed in fp64 and rounded once"""
    k, n, m = 1024, 320, 270
    g = torch.Generator().manual_seed(24)
    _, q_w, _, _ = problem(24, 1, k, n, torch.bfloat16)
    s = (2.0 ** torch.randint(-6, 0, (k // 128, n), generator=g).float()).to(torch.bfloat16)   # power-of-two scales
    a = torch.randint(-3, 4, (m, k), generator=g).to(torch.bfloat16)
    out = native_gemm(a, q_w, s, k, n, gpu_device).cpu()
    w = (q_w.float() - 8) * s.float().repeat_interleave(128, 0)
    want = (a.double() @ w.double()).to(torch.bfloat16)
    assert torch.equal(out.view(torch.int16), want.view(torch.int16))


@pytest.mark.parametrize("m", [65, 256, 384])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_prefill_silu_mul_epilogue_matches_separate_ops(gpu_device, prefill, m, dtype):
    from neural_magic_vllm_amd import _custom_ops as ops
    from neural_magic_vllm_amd.model_executor.layers.quantization.gptq_marlin import GPTQMarlinLinearMethod as LM
    k, inter = 512, 1024
    n = 2 * inter
    a, q_w, s, _ = problem(25, m, k, n, dtype)
    q_i, s_i = LM._interleave_gate_up(q_w), LM._interleave_gate_up(s)
    prefill.setenv("NMV_W4P_SPLITS", "1")   # the fused launch never splits K; a split plain launch sums its slabs in another order
    fused = native_gemm(a, q_i, s_i, k, n, gpu_device, 1)
    plain = native_gemm(a, q_w, s, k, n, gpu_device, 0)
    want = torch.empty((m, inter), dtype=dtype, device=gpu_device)
    ops.silu_and_mul(want, plain)
    assert fused.shape == (m, inter)
    assert torch.equal(fused.view(torch.int16), want.view(torch.int16))


def test_prefill_gemm_rows_past_m_do_not_leak(gpu_device, prefill):
    """rows >= M of the 256-row tile are never fetched (clamped DMA) nor stored: memory behind C stays put, and whatever
    lies behind A has no influence"""
    from neural_magic_vllm_amd import _custom_ops as ops
    k, n, m = 1024, 256, 300
    a, q_w, s, w_ref = problem(27, 512, k, n, torch.bfloat16)
    d = gpu_device
    b = ops.w4_native_repack(ref_math.gptq_pack(q_w, 4, k, n).to(d), None, k, n)
    ws = torch.zeros(64, dtype=torch.int32, device=d)
    big_a = a.to(d)
    big_a[m:] = float("nan")
    out = ops.w4_native_gemm(big_a[:m], b, s.to(d), ws, m, n, k, 0)
    assert torch.isfinite(out.float()).all()
    assert rel_err(out.cpu(), (a[:m].double() @ w_ref.double()).float()) < 6e-3


@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
@pytest.mark.parametrize("m,k,n", [(100, 512, 448), (512, 4096, 1024), (300, 1024, 6144)])
def test_prefill_gemm_slabs_in_the_model_dtype(gpu_device, prefill, dtype, m, k, n):
    """mode 3: the deferred slabs in the model dtype are the fp32 slabs of mode 2, each element rounded once"""
    from neural_magic_vllm_amd import _custom_ops as ops
    assert ops.w4_native_gemm_slab16(m, n, k) and not ops.w4_native_gemm_slab16(64, n, k)
    a, q_w, s, w_ref = problem(31, m, k, n, dtype)
    s32 = native_gemm(a, q_w, s, k, n, gpu_device, 2)
    s16 = native_gemm(a, q_w, s, k, n, gpu_device, 3)
    assert s16.dtype == dtype and s16.shape == s32.shape
    assert torch.equal(s16.view(torch.int16), s32.to(dtype).view(torch.int16))
    out = s16.float().sum(0)
    assert rel_err(out.cpu(), (a.double() @ w_ref.double()).float()) < 6e-3


@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_slab_consumers_read_model_dtype_slabs(gpu_device, dtype):
    """the norm and the rope + cache launches on slabs in the model dtype: bit-identical to the same launches on fp32
    slabs that hold the same (already rounded) values"""
    from neural_magic_vllm_amd import _custom_ops as ops
    d = gpu_device
    g = torch.Generator().manual_seed(41)
    splits, t, hidden = 5, 37, 4096
    slab16 = (torch.randn((splits, t, hidden), generator=g) * 0.5).to(dtype).to(d)
    res = torch.randn((t, hidden), generator=g).to(dtype).to(d)
    w = torch.randn((hidden, ), generator=g).to(dtype).to(d)
    r16, r32 = res.clone(), res.clone()
    o16 = ops.fused_add_rms_norm_partial(slab16, r16, w, 1e-5)
    o32 = ops.fused_add_rms_norm_partial(slab16.float(), r32, w, 1e-5)
    assert torch.equal(o16.view(torch.int16), o32.view(torch.int16)) and torch.equal(r16.view(torch.int16), r32.view(torch.int16))
    heads, kv_heads, hs, bs = 8, 2, 128, 16
    nq = (heads + 2 * kv_heads) * hs
    qkv16 = (torch.randn((3, t, nq), generator=g)).to(dtype).to(d)
    pos = torch.randint(0, 500, (t, ), generator=g).to(d)
    cos_sin = torch.randn((512, hs), generator=g).to(dtype).to(d)
    slots = torch.randperm(8 * bs, generator=g)[:t].to(d)
    outs = []
    for sl in (qkv16, qkv16.float()):
        kc = torch.zeros((8, kv_heads, hs // 8, bs, 8), dtype=dtype, device=d)
        vc = torch.zeros((8, kv_heads, hs, bs), dtype=dtype, device=d)
        q = ops.rotary_embedding_and_cache_partial(pos, sl, heads, kv_heads, hs, cos_sin, kc, vc, slots, "auto", 1.0, dtype)
        outs.append((q, kc, vc))
    for x, y in zip(outs[0], outs[1]):
        assert torch.equal(x.view(torch.int16), y.view(torch.int16))
