"""GPU parity: scaled_int8_quant, scaled_fp8_quant and cutlass_scaled_mm (int8 / fp8) vs the oracle.
Recipes and tolerances: the reference's tests/kernels/test_int8_quant.py:25-71 (atol 1),
tests/kernels/test_cutlass.py:51-140 (int8 rtol 1e-1 atol 1; fp8 rtol 1e-2 atol 5e-2).  The int8
GEMM accumulates exactly in int32, so it is additionally compared tightly."""
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("num_tokens", [1, 7, 83])
@pytest.mark.parametrize("hidden", [16, 67, 768, 5120, 8192])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_dynamic_scaled_int8_quant(gpu_device, num_tokens, hidden, dtype):
    from neural_magic_vllm_amd import _custom_ops as ops
    g = torch.Generator().manual_seed(0)
    x = (torch.rand((num_tokens, hidden), generator=g) * 1000).to(dtype)
    out, scales = ops.scaled_int8_quant(x.to(gpu_device))
    ref, ref_scales = oracle.scaled_int8_quant(x)
    assert torch.allclose(scales.cpu(), ref_scales, rtol=1e-6, atol=0)
    assert torch.equal(out.cpu(), ref)  # same fp32 expression, round-to-nearest-even: exact
    # the reference test's own checker
    xf = x.float()
    chk = (xf * (127.0 / xf.abs().max(dim=-1, keepdim=True)[0])).round().clamp(-128, 127).to(torch.int8)
    assert torch.allclose(out.cpu(), chk, atol=1)


@pytest.mark.parametrize("num_tokens", [1, 83])
@pytest.mark.parametrize("hidden", [67, 5120])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
@pytest.mark.parametrize("scale", [0.1, 0.5, 0.8, 1.2, 2.1])
def test_static_scaled_int8_quant(gpu_device, num_tokens, hidden, dtype, scale):
    from neural_magic_vllm_amd import _custom_ops as ops
    g = torch.Generator().manual_seed(0)
    x = (torch.rand((num_tokens, hidden), generator=g) * 1000 - 300).to(dtype)
    s = torch.tensor([scale], dtype=torch.float32)
    out, _ = ops.scaled_int8_quant(x.to(gpu_device), s.to(gpu_device))
    ref, _ = oracle.scaled_int8_quant(x, s)
    assert torch.equal(out.cpu(), ref)
    chk = (x.float() / scale).round().clamp(-128, 127).to(torch.int8)
    assert torch.allclose(out.cpu(), chk, atol=1)


@pytest.mark.parametrize("shape", [(1, 128), (33, 4096), (7, 8199), (512, 1024)])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
@pytest.mark.parametrize("dynamic", [True, False])
def test_scaled_fp8_quant(gpu_device, shape, dtype, dynamic):
    from neural_magic_vllm_amd import _custom_ops as ops
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(shape, generator=g) * 3).to(dtype)
    if dynamic:
        out, scale = ops.scaled_fp8_quant(x.to(gpu_device))
        ref, ref_scale = oracle.scaled_fp8_quant(x)
    else:
        s = torch.tensor([0.07], dtype=torch.float32)
        out, scale = ops.scaled_fp8_quant(x.to(gpu_device), s.to(gpu_device))
        ref, ref_scale = oracle.scaled_fp8_quant(x, s)
    assert out.dtype == torch.float8_e4m3fn
    assert torch.equal(scale.cpu(), ref_scale)
    assert torch.equal(out.cpu().view(torch.uint8), ref)  # bytes, bit for bit
    # the reference's per_tensor_quantize semantics (fp8.py:601-605)
    chk = (x.float() / ref_scale).clamp(-448, 448).to(torch.float8_e4m3fn)
    diff = (out.cpu().float() - chk.float()).abs()
    assert (diff <= chk.float().abs() * 0.13 + 1e-3).all()  # 1/scale vs /scale: at most 1 fp8 ulp


def test_scaled_fp8_quant_padding(gpu_device):
    from neural_magic_vllm_amd import _custom_ops as ops
    x = torch.randn((5, 256), dtype=torch.bfloat16, device=gpu_device)
    out, scale = ops.scaled_fp8_quant(x, batch_dim_padding=17)
    assert out.shape == (17, 256)
    ref, _ = oracle.scaled_fp8_quant(x.cpu())
    assert torch.equal(out[:5].cpu().view(torch.uint8), ref)


def to_int8(t):
    return torch.round(t.clamp(min=-128, max=127)).to(dtype=torch.int8)


def to_fp8(t):
    return torch.round(t.clamp(min=-448, max=448)).to(dtype=torch.float8_e4m3fn)


@pytest.mark.parametrize("m", [512, 222, 100, 33, 1])
@pytest.mark.parametrize("n", [2048, 256, 1024])
@pytest.mark.parametrize("k", [128, 496, 1024])
@pytest.mark.parametrize("per_act_token", [True, False])
@pytest.mark.parametrize("per_out_ch", [True, False])
@pytest.mark.parametrize("use_bias", [True, False])
@pytest.mark.parametrize("kind", ["int8", "fp8"])
def test_cutlass_scaled_mm(gpu_device, m, n, k, per_act_token, per_out_ch, use_bias, kind):
    from neural_magic_vllm_amd import _custom_ops as ops
    g = torch.Generator().manual_seed(0)
    out_dtype = torch.bfloat16
    conv = to_int8 if kind == "int8" else to_fp8
    a = conv(torch.randn((m, k), generator=g) * (5 if kind == "int8" else 1))
    b = conv(torch.randn((n, k), generator=g) * (5 if kind == "int8" else 1)).t()  # column-major [k, n]
    scale_a = torch.randn((m if per_act_token else 1, 1), generator=g).abs() / 10
    scale_b = torch.randn((1, n if per_out_ch else 1), generator=g).abs() / 10
    bias = (torch.rand((n, ), generator=g) * 10).to(out_dtype) if use_bias else None
    d = gpu_device
    out = ops.cutlass_scaled_mm(a.to(d), b.t().contiguous().to(d).t(), scale_a.to(d), scale_b.to(d),
                                out_dtype, None if bias is None else bias.to(d)).cpu()
    ref = oracle.scaled_mm(a, b, scale_a, scale_b, out_dtype, bias)
    baseline = (scale_a * (scale_b * (a.float() @ b.float()))).to(out_dtype)
    if bias is not None:
        baseline = baseline + bias
    if kind == "int8":
        assert torch.allclose(out.float(), baseline.float(), rtol=1e-1, atol=1e0)
        # exact int32 accumulation + one fp32 epilogue rounding: within 1 bf16 ulp of the oracle
        assert torch.allclose(out.float(), ref.float(), rtol=2**-7, atol=1e-3)
    else:
        assert torch.allclose(out.float(), baseline.float(), rtol=1e-2, atol=5e-2)
        assert torch.allclose(out.float(), ref.float(), rtol=2**-7, atol=5e-3)


@pytest.mark.parametrize("out_dtype", [torch.half, torch.bfloat16])
@pytest.mark.parametrize("kind", ["int8", "fp8"])
def test_cutlass_scaled_mm_llama_decode_shapes(gpu_device, out_dtype, kind):
    """config 4 of BASELINE.json: Llama-3-8B w8a8 decode, M = 16, the o_proj / down_proj shapes"""
    from neural_magic_vllm_amd import _custom_ops as ops
    g = torch.Generator().manual_seed(1)
    conv = to_int8 if kind == "int8" else to_fp8
    for k, n in ((4096, 4096), (14336, 4096)):
        a = conv(torch.randn((16, k), generator=g) * 20)
        bt = conv(torch.randn((n, k), generator=g) * 20)
        sa = torch.rand((16, 1), generator=g) / 100
        sb = torch.rand((1, n), generator=g) / 100
        d = gpu_device
        out = ops.cutlass_scaled_mm(a.to(d), bt.to(d).t(), sa.to(d), sb.to(d), out_dtype).cpu()
        ref = oracle.scaled_mm(a, bt.t(), sa, sb, out_dtype)
        tol = 2**-10 if out_dtype == torch.half else 2**-7
        assert torch.allclose(out.float(), ref.float(), rtol=tol * 1.01, atol=1e-3 if kind == "int8" else 2e-2)


def test_cutlass_scaled_mm_strided_output_and_checks(gpu_device):
    from neural_magic_vllm_amd import _custom_ops as ops
    d = gpu_device
    g = torch.Generator().manual_seed(2)
    m, n, k = 37, 256, 512
    a = to_int8(torch.randn((m, k), generator=g) * 5)
    bt = to_int8(torch.randn((n, k), generator=g) * 5)
    sa, sb = torch.rand((m, 1), generator=g), torch.rand((1, n), generator=g)
    big = torch.zeros((m, 2 * n), dtype=torch.bfloat16, device=d)
    out = big[:, :n]  # row stride 2n (a multiple of 16), as the reference's stride test
    torch.ops._C.cutlass_scaled_mm(out, a.to(d), bt.to(d).t(), sa.to(d), sb.to(d), None)
    ref = oracle.scaled_mm(a, bt.t(), sa, sb, torch.bfloat16)
    assert torch.allclose(out.cpu().float(), ref.float(), rtol=2**-7, atol=1e-3)
    assert float(big[:, n:].abs().sum()) == 0
    with pytest.raises(RuntimeError, match="column major"):
        torch.ops._C.cutlass_scaled_mm(torch.empty((m, n), dtype=torch.bfloat16, device=d), a.to(d),
                                       bt.t().contiguous().to(d), sa.to(d), sb.to(d), None)
    assert ops.cutlass_scaled_mm_supports_fp8(95) and not ops.cutlass_scaled_mm_supports_fp8(80)


@pytest.mark.parametrize("kind", ["int8", "fp8"])
@pytest.mark.parametrize("m,n,k", [(64, 4096, 4096), (512, 6144, 4096), (130, 208, 14336), (48, 4096, 14336),
                                   (300, 1008, 2064), (2048, 512, 1024), (64, 4096, 14336)])
def test_cutlass_scaled_mm_split_k_and_tiles(gpu_device, monkeypatch, kind, m, n, k):
    """M > 32: the per-wave column-tile count NT and the split-K slab count are chosen by shape.
    int8 accumulates exactly in int32 whatever the split, so every variant is bit-identical; fp8
    differs in fp32 summation order only.  All are checked against the oracle as well."""
    from neural_magic_vllm_amd import _custom_ops as ops
    g = torch.Generator().manual_seed(3)
    conv = to_int8 if kind == "int8" else to_fp8
    a = conv(torch.randn((m, k), generator=g) * (20 if kind == "int8" else 1))
    bt = conv(torch.randn((n, k), generator=g) * (20 if kind == "int8" else 1))
    sa = torch.rand((m, 1), generator=g) / 100
    sb = torch.rand((1, n), generator=g) / 100
    bias = torch.randn((n, ), generator=g).to(torch.bfloat16)
    d = gpu_device
    args = (a.to(d), bt.to(d).t(), sa.to(d), sb.to(d), torch.bfloat16, bias.to(d))
    outs = [ops.cutlass_scaled_mm(*args).cpu()]
    for nt, splits in ((1, 1), (2, 3), (4, 4), (4, 1)):
        monkeypatch.setenv("NMV_MM_NT", str(nt))
        monkeypatch.setenv("NMV_MM_SPLITS", str(splits))
        outs.append(ops.cutlass_scaled_mm(*args).cpu())
    ref = oracle.scaled_mm(a, bt.t(), sa, sb, torch.bfloat16, bias)
    if kind == "int8":
        for i, o in enumerate(outs[1:]):
            assert torch.equal(outs[0].view(torch.int16), o.view(torch.int16)), i
        assert torch.allclose(outs[0].float(), ref.float(), rtol=2**-7 * 1.01, atol=1e-3)
    else:
        for o in outs:
            assert torch.allclose(o.float(), ref.float(), rtol=2**-7 * 1.01, atol=2e-2)


@pytest.mark.parametrize("kind", ["int8", "fp8"])
@pytest.mark.parametrize("m,n,k", [(17, 256, 512), (33, 208, 1024), (48, 6144, 4096), (64, 4096, 14336), (64, 1088, 2048)])
def test_cutlass_scaled_mm_wide_kernel(gpu_device, monkeypatch, kind, m, n, k):
    """17 .. 64 rows, K in whole 256-byte chunks: the waves of a workgroup split N and share the activations through LDS
    (scaled_mm_wide_kernel).  Against the K-splitting kernel (NMV_MM_WIDE=0): bit-identical for int8 (exact int32
    sums), fp32 summation order apart for fp8; every column-tile count and slice count; ragged N; the oracle."""
    from neural_magic_vllm_amd import _custom_ops as ops
    g = torch.Generator().manual_seed(5)
    conv = to_int8 if kind == "int8" else to_fp8
    a = conv(torch.randn((m, k), generator=g) * (20 if kind == "int8" else 1))
    bt = conv(torch.randn((n, k), generator=g) * (20 if kind == "int8" else 1))
    sa = torch.rand((m, 1), generator=g) / 100
    sb = torch.rand((1, n), generator=g) / 100
    bias = torch.randn((n, ), generator=g).to(torch.bfloat16)
    d = gpu_device
    args = (a.to(d), bt.to(d).t(), sa.to(d), sb.to(d), torch.bfloat16, bias.to(d))
    monkeypatch.setenv("NMV_MM_WIDE", "0")
    old = ops.cutlass_scaled_mm(*args).cpu()
    monkeypatch.setenv("NMV_MM_WIDE", "1")
    outs = [ops.cutlass_scaled_mm(*args).cpu()]
    for nt, splits in ((1, 1), (2, 2), (4, 1), (4, 4), (1, 8)):
        monkeypatch.setenv("NMV_MM_NT", str(nt))
        monkeypatch.setenv("NMV_MM_SPLITS", str(splits))
        outs.append(ops.cutlass_scaled_mm(*args).cpu())
    ref = oracle.scaled_mm(a, bt.t(), sa, sb, torch.bfloat16, bias)
    for i, o in enumerate(outs):
        if kind == "int8":
            assert torch.equal(old.view(torch.int16), o.view(torch.int16)), i
        assert torch.allclose(o.float(), ref.float(), rtol=2**-7 * 1.01, atol=1e-3 if kind == "int8" else 2e-2), i


# ---------------------------------------------------------------------------------------------
# against fixtures produced by the REFERENCE's own code (tools/make_golden_w8a8.py): its test helpers'
# inputs + baseline_scaled_mm outputs, its int8-quant test's expected tensors, per_tensor_quantize
import glob  # noqa: E402
import os  # noqa: E402

import numpy as np  # noqa: E402

import helpers  # noqa: E402
from test_oracle_golden_w8a8 import DT, GOLD, MM, load_mm, mm_close  # noqa: E402


@pytest.mark.parametrize("name", MM)
def test_cutlass_scaled_mm_vs_reference_baseline(gpu_device, name):
    from neural_magic_vllm_amd import _custom_ops as ops
    c = load_mm(name)
    a = c["a"].to(gpu_device)
    b = c["b"].t().contiguous().to(gpu_device).t()          # [K, N] column-major, as the reference passes it
    out = ops.cutlass_scaled_mm(a, b, c["scale_a"].to(gpu_device), c["scale_b"].to(gpu_device), c["dtype"],
                                None if c["bias"] is None else c["bias"].to(gpu_device))
    assert mm_close(c["kind"], out.cpu().float(), c["baseline"].float()), name      # test_cutlass.py:79 / :110
    ref = oracle.scaled_mm(c["a"], c["b"], c["scale_a"], c["scale_b"], c["dtype"], c["bias"])
    if c["kind"] == "int8":
        # int32 accumulation is exact and the epilogue is the same fp32 expression: the bits agree, except that
        # the two compilers may round the fp32 intermediate of an exact tie of the output type differently
        # (seen: 1 element of 56 832, one output ulp apart)
        o16, r16 = out.cpu().view(torch.int16).int(), ref.view(torch.int16).int()
        diff = (o16 - r16).abs()
        assert int(diff.max()) <= 1 and float((diff != 0).float().mean()) < 1e-4, name


def test_scaled_int8_quant_vs_reference_expectation(gpu_device):
    from neural_magic_vllm_amd import _custom_ops as ops
    g = np.load(os.path.join(GOLD, "int8_quant.npz"))
    for j in range(int(g["n_dynamic"])):
        x = helpers.from_np(g[f"dyn{j}_x"], DT[str(g[f"dyn{j}_dtype"])])
        q, s = ops.scaled_int8_quant(x.to(gpu_device))
        assert torch.allclose(s.cpu(), torch.from_numpy(g[f"dyn{j}_s"]))               # test_int8_quant.py:44
        assert torch.allclose(q.cpu(), torch.from_numpy(g[f"dyn{j}_q"]), atol=1)       # :45-46
    for j in range(int(g["n_static"])):
        x = helpers.from_np(g[f"sta{j}_x"], DT[str(g[f"sta{j}_dtype"])])
        sc = torch.tensor([float(g[f"sta{j}_scale"])], dtype=torch.float32, device=gpu_device)
        q, _ = ops.scaled_int8_quant(x.to(gpu_device), sc)
        assert torch.allclose(q.cpu(), torch.from_numpy(g[f"sta{j}_q"]), atol=1)       # :69-71


def test_static_scaled_fp8_quant_vs_reference_per_tensor_quantize(gpu_device):
    from neural_magic_vllm_amd import _custom_ops as ops
    g = np.load(os.path.join(GOLD, "fp8_quant.npz"))
    for j in range(int(g["n_ptq"])):
        x = helpers.from_np(g[f"ptq{j}_x"], DT[str(g[f"ptq{j}_dtype"])])
        inv = torch.tensor([float(g[f"ptq{j}_inv_scale"])], dtype=torch.float32, device=gpu_device)
        q, _ = ops.scaled_fp8_quant(x.to(gpu_device), inv)
        got = q.cpu().view(torch.uint8)
        want = torch.from_numpy(g[f"ptq{j}_q"])
        diff = (got.int() - want.int()).abs()
        # fp8.py:601-605 divides in the tensor dtype, the kernel in fp32 (fp8/common.cu:29): equal bytes except
        # at ties of that intermediate rounding
        assert int(diff.max()) <= 1 and float((diff != 0).float().mean()) < 0.05, j
        ref, _ = oracle.scaled_fp8_quant(x, inv.cpu())
        assert torch.equal(got, ref)
