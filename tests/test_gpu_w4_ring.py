"""GPU: the loader / consumer ring kernel (csrc/w4a16_ring.hip) behind nmv_w4_native_gemm for calls of 17..64 rows,
group 128.  The arithmetic is the stream kernel's (fp32 group scaling, zero point as an MFMA k-step), so the checks
are those of test_gpu_w4_native.py, forced onto this kernel at every row count it accepts: the oracle's a @ w_ref at
the reference's tolerance (test_marlin_gemm.py:172-179, < 0.04) and this repo's own (< 6e-3), a one-hot row exact,
the three modes consistent with each other, run-to-run bit-identity, ragged strips (N % 128 == 64), every split-K
count, and no workgroup may have given up on a ring slot (nmv_w4_ring_timeouts)."""
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
    ws = torch.zeros(max(n // 64 * 16, 16), dtype=torch.int32, device=dev)
    out = ops.w4_native_gemm(a.to(dev), b, s.to(dev), ws, a.shape[0], n, k, mode)
    assert int(ws.abs().sum()) == 0, "the ticket array must be returned zeroed"
    return out


@pytest.fixture()
def ring(monkeypatch):
    """every call of 17..64 rows takes the ring kernel; afterwards no slot wait may have timed out"""
    from neural_magic_vllm_amd import _lib
    monkeypatch.setenv("NMV_W4R", "1")
    monkeypatch.setenv("NMV_W4R_MIN_M", "17")
    monkeypatch.setenv("NMV_W4R_MIN_WGS", "1")
    monkeypatch.setenv("NMV_W4R_PREFILL", "1")
    monkeypatch.setenv("NMV_W4P", "0")   # prompt-sized calls: this kernel's 128-row variant, not csrc/w4a16_prefill.hip
    yield monkeypatch
    assert _lib.load().nmv_w4_ring_timeouts() == 0


def test_ring_plan_is_taken(gpu_device, ring):
    """the splits query answers from the ring plan in its row range (Llama-3-8B o_proj: more than one slab)"""
    from neural_magic_vllm_amd import _lib
    lib = _lib.load()
    ring.setenv("NMV_W4R_SPLITS", "4")
    assert lib.nmv_w4_native_gemm_splits(64, 4096, 4096, 32) == 4
    ring.setenv("NMV_W4R_SPLITS", "2")
    assert lib.nmv_w4_native_gemm_splits(40, 4096, 4096, 32) == 2


@pytest.mark.parametrize("k,n", [(256, 64), (384, 128), (1024, 448), (2048, 1088), (4096, 6144), (14336, 4096)])
@pytest.mark.parametrize("m", [17, 31, 32, 33, 47, 48, 49, 63, 64])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_ring_gemm(gpu_device, ring, k, n, m, dtype):
    a, q_w, s, w_ref = problem(1, m, k, n, dtype)
    out = native_gemm(a, q_w, s, k, n, gpu_device).cpu()
    ref = (a.double() @ w_ref.double()).float()
    e = rel_err(out, ref)
    assert e < 0.04 and e < 6e-3, e
    assert (out.float() - ref).abs().max().item() < 0.04 * ref.abs().max().item() + 1e-2


@pytest.mark.parametrize("k,n", [(256, 64), (1024, 448), (4096, 6144), (14336, 4096)])
@pytest.mark.parametrize("m", [65, 128, 129, 200, 512])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_ring_gemm_prompt_sized(gpu_device, ring, k, n, m, dtype):
    """65 rows and more: row blocks of the 128-row tile (one k-lane, activation fragments read k-step by k-step, row
    tile cb summed by column-block wave cb)"""
    a, q_w, s, w_ref = problem(11, m, k, n, dtype)
    out = native_gemm(a, q_w, s, k, n, gpu_device).cpu()
    ref = (a.double() @ w_ref.double()).float()
    e = rel_err(out, ref)
    assert e < 0.04 and e < 6e-3, e
    assert (out.float() - ref).abs().max().item() < 0.04 * ref.abs().max().item() + 1e-2


def test_ring_gemm_prompt_sized_one_hot_and_modes(gpu_device, ring):
    k, n, m = 512, 448, 150
    _, q_w, s, _ = problem(12, 1, k, n, torch.bfloat16)
    rows = [(37 * i) % k for i in range(m)]
    a = torch.zeros((m, k), dtype=torch.bfloat16)
    for i, r in enumerate(rows):
        a[i, r] = 1.0
    out = native_gemm(a, q_w, s, k, n, gpu_device)
    want = ((q_w[rows].float() - 8) * s.float()[[r // 128 for r in rows]]).to(torch.bfloat16)
    assert torch.equal(out.cpu().view(torch.int16), want.view(torch.int16))
    slab = native_gemm(a, q_w, s, k, n, gpu_device, 2)
    acc = torch.zeros_like(slab[0])
    for i in range(slab.shape[0]):
        acc = acc + slab[i]
    assert torch.equal(acc.to(torch.bfloat16).view(torch.int16), out.view(torch.int16))


@pytest.mark.parametrize("splits", [1, 2, 4, 8, 16])
@pytest.mark.parametrize("m", [20, 64])
def test_ring_gemm_every_split_count(gpu_device, ring, splits, m):
    ring.setenv("NMV_W4R_SPLITS", str(splits))
    ring.setenv("NMV_W4R_MAX_SPLITS", "64")
    k, n = 4096, 576
    a, q_w, s, w_ref = problem(2, m, k, n, torch.bfloat16)
    out = native_gemm(a, q_w, s, k, n, gpu_device)
    assert rel_err(out.cpu(), (a.double() @ w_ref.double()).float()) < 6e-3
    slab = native_gemm(a, q_w, s, k, n, gpu_device, 2)
    assert slab.shape[0] == splits
    acc = torch.zeros_like(slab[0])
    for i in range(splits):
        acc = acc + slab[i]
    assert torch.equal(acc.to(torch.bfloat16).view(torch.int16), out.view(torch.int16))


@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_ring_gemm_one_hot_is_exact_dequant(gpu_device, ring, dtype):
    """a one-hot activation row selects one weight row: the output must be round((q - 8) * s) exactly"""
    k, n = 512, 448
    _, q_w, s, _ = problem(3, 1, k, n, dtype)
    rows = [0, 1, 7, 8, 31, 32, 63, 64, 95, 127, 128, 129, 255, 256, 300, 383, 384, 511] + list(range(40, 60))
    a = torch.zeros((len(rows), k), dtype=dtype)
    for i, r in enumerate(rows):
        a[i, r] = 1.0
    out = native_gemm(a, q_w, s, k, n, gpu_device).cpu()
    want = ((q_w[rows].float() - 8) * s.float()[[r // 128 for r in rows]]).to(dtype)
    assert torch.equal(out.view(torch.int16), want.view(torch.int16))


@pytest.mark.parametrize("m", [24, 64])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_ring_silu_mul_epilogue_matches_separate_ops(gpu_device, ring, m, dtype):
    from neural_magic_vllm_amd import _custom_ops as ops
    from neural_magic_vllm_amd.model_executor.layers.quantization.gptq_marlin import GPTQMarlinLinearMethod as LM
    k, inter = 512, 8192
    n = 2 * inter
    a, q_w, s, _ = problem(5, m, k, n, dtype)
    ring.setenv("NMV_W4R_SPLITS", "1")
    plain = native_gemm(a, q_w, s, k, n, gpu_device, 0)
    want = torch.empty((m, inter), dtype=dtype, device=gpu_device)
    ops.silu_and_mul(want, plain)
    qi, si = LM._interleave_gate_up(q_w), LM._interleave_gate_up(s)
    got = native_gemm(a, qi, si, k, n, gpu_device, 1)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))


@pytest.mark.parametrize("m", [33, 64])
def test_ring_gemm_full_gate_up_twice(gpu_device, ring, m):
    """the whole Llama-3-8B gate_up projection: every column against dequantise-then-matmul, and two launches bit-equal"""
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
    for c0 in range(0, n, 4096):
        w = ((q_w[:, c0:c0 + 4096].float() - 8) * s.float()[:, c0:c0 + 4096].repeat_interleave(128, 0)).double()
        ref = (a.double() @ w).float()
        assert rel_err(c[:, c0:c0 + 4096], ref) < 6e-3, c0


def test_ring_gemm_rows_past_m_do_not_leak(gpu_device, ring):
    """rows >= M of the 16 MT-row tile are never fetched (bounds-checked DMA) nor stored: memory behind C stays put"""
    from neural_magic_vllm_amd import _custom_ops as ops
    k, n, m = 1024, 256, 35
    a, q_w, s, w_ref = problem(7, 64, k, n, torch.bfloat16)
    d = gpu_device
    b = ops.w4_native_repack(ref_math.gptq_pack(q_w, 4, k, n).to(d), None, k, n)
    ws = torch.zeros(64, dtype=torch.int32, device=d)
    big = a.to(d)
    big[m:] = float("nan")          # rows past M are poison: they must not reach the first M rows of C
    out = ops.w4_native_gemm(big[:m], b, s.to(d), ws, m, n, k, 0)
    assert torch.isfinite(out.float()).all()
    assert rel_err(out.cpu(), (a[:m].double() @ w_ref.double()).float()) < 6e-3
