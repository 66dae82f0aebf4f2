"""CPU: the W8A8 leg of the oracle (oracle.c scaled_mm / scaled_int8_quant / scaled_fp8_quant / fp8 codec),
the v1/v2 heuristic and the LinearMethod parameter tables against fixtures that tools/make_golden_w8a8.py
produced from the REFERENCE's own code (its test_cutlass.py helpers and baseline_scaled_mm, its
test_int8_quant.py expectations, fp8.py per_tensor_quantize, marlin_utils.pack_fp8_to_int32,
PagedAttention.forward_decode, the LinearMethod classes)."""
import glob
import json
import os

import numpy as np
import pytest
import torch

import helpers
import oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DT = {"bfloat16": torch.bfloat16, "float16": torch.float16}
MM = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "scaled_mm_*.npz")))


def load_mm(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    kind, dt = str(g["kind"]), DT[str(g["out_dtype"])]
    a = torch.from_numpy(g["a_bytes"].copy())
    bt = torch.from_numpy(g["b_t_bytes"].copy())             # [N, K] row-major
    if kind == "fp8":
        a, bt = a.view(torch.float8_e4m3fn), bt.view(torch.float8_e4m3fn)
    else:
        a, bt = a.view(torch.int8), bt.view(torch.int8)
    bias = helpers.from_np(g["bias"], dt) if bool(g["use_bias"]) else None
    return dict(kind=kind, dtype=dt, a=a, b=bt.t(), scale_a=torch.from_numpy(g["scale_a"].copy()),
                scale_b=torch.from_numpy(g["scale_b"].copy()), bias=bias, baseline=helpers.from_np(g["baseline"], dt))


def mm_close(kind, out, baseline):
    """the reference's own acceptance (test_cutlass.py:79 fp8, :110 int8)"""
    if kind == "fp8":
        return torch.allclose(out, baseline, rtol=1e-2, atol=5e-2)
    return torch.allclose(out, baseline, rtol=1e-1, atol=1e0)


def test_fixture_set_is_complete():
    assert len(MM) >= 6 and any("fp8" in m for m in MM) and any("int8" in m for m in MM)


@pytest.mark.parametrize("name", MM)
def test_scaled_mm_oracle_vs_reference_baseline(name):
    c = load_mm(name)
    out = oracle.scaled_mm(c["a"], c["b"], c["scale_a"], c["scale_b"], c["dtype"], c["bias"])
    assert mm_close(c["kind"], out.float(), c["baseline"].float()), name
    if c["kind"] == "int8" and c["bias"] is None:
        # exact int32 accumulation + fp32 epilogue: at most one rounding step of the output dtype apart from
        # baseline_scaled_mm's fp32 matmul
        ulp = (c["baseline"].float().abs() * (2.0**-7 if c["dtype"] == torch.bfloat16 else 2.0**-10)).clamp_min(1e-3)
        assert ((out.float() - c["baseline"].float()).abs() <= 2 * ulp).all(), name


def test_int8_quant_oracle_vs_reference_expectation():
    g = np.load(os.path.join(GOLD, "int8_quant.npz"))
    for j in range(int(g["n_dynamic"])):
        x = helpers.from_np(g[f"dyn{j}_x"], DT[str(g[f"dyn{j}_dtype"])])
        q, s = oracle.scaled_int8_quant(x)
        assert torch.allclose(s, torch.from_numpy(g[f"dyn{j}_s"]))                      # test_int8_quant.py:44
        assert torch.allclose(q, torch.from_numpy(g[f"dyn{j}_q"]), atol=1)              # :45-46
        # the fixture's values are rounded by torch, the kernel semantics by rintf: same tie rule -> exact,
        # except where x / scale itself differs in the last place
        assert (q.int() - torch.from_numpy(g[f"dyn{j}_q"]).int()).abs().float().mean() < 0.02
    for j in range(int(g["n_static"])):
        x = helpers.from_np(g[f"sta{j}_x"], DT[str(g[f"sta{j}_dtype"])])
        q, _ = oracle.scaled_int8_quant(x, torch.tensor([float(g[f"sta{j}_scale"])], dtype=torch.float32))
        assert torch.allclose(q, torch.from_numpy(g[f"sta{j}_q"]), atol=1)              # :69-71


def test_fp8_quant_and_pack_oracle_vs_reference():
    g = np.load(os.path.join(GOLD, "fp8_quant.npz"))
    for j in range(int(g["n_ptq"])):
        x = helpers.from_np(g[f"ptq{j}_x"], DT[str(g[f"ptq{j}_dtype"])])
        inv = torch.tensor([float(g[f"ptq{j}_inv_scale"])], dtype=torch.float32)
        q, _ = oracle.scaled_fp8_quant(x, inv)
        want = torch.from_numpy(g[f"ptq{j}_q"])
        # per_tensor_quantize divides in the tensor's dtype, the kernel (fp8/common.cu:29) in fp32: equal bytes
        # except at rounding ties of that intermediate -- at most one fp8 code apart, and rarely
        diff = (q.int() - want.int()).abs()
        assert int(diff.max()) <= 1 and float((diff != 0).float().mean()) < 0.05, j
        assert int(q[0, 0]) == int(want[0, 0]) == 0x7E and int(q[0, 1]) == int(want[0, 1]) == 0xFE   # +-448
    # the fp8 byte codec itself: decode(encode) is the identity on every code that is not NaN
    codes = torch.arange(256, dtype=torch.uint8)
    vals = oracle.fp8_decode(codes)
    ok = ~torch.isnan(vals)
    assert torch.equal(oracle.fp8_encode(vals[ok]), codes[ok])
    assert torch.equal(vals[ok], codes[ok].view(torch.float8_e4m3fn).float())
    # pack_fp8_to_int32 (marlin_utils.py:227-247): 4 consecutive-K bytes per int32, little end first
    from oracle import ref_math
    w = torch.from_numpy(g["pack_in"].copy())
    assert torch.equal(ref_math.pack_fp8_to_int32(w), torch.from_numpy(g["pack_out"]))


def test_paged_attention_v1_v2_truth_table():
    """every row of the reference's own choice and tmp-buffer shapes (paged_attn.py:112-153)"""
    from neural_magic_vllm_amd.attention.ops.paged_attn import PagedAttention, _PARTITION_SIZE
    t = json.load(open(os.path.join(GOLD, "pa_heuristic.json")))
    assert len(t["table"]) >= 150
    for row in t["table"]:
        ns, nh, msl = row["num_seqs"], row["num_heads"], row["max_seq_len"]
        assert PagedAttention.use_v1(msl, ns, nh) == row["use_v1"], row
        if not row["use_v1"]:
            parts = (msl + _PARTITION_SIZE - 1) // _PARTITION_SIZE
            assert parts == row["partitions"] and row["tmp_shape"] == [ns, nh, parts, t["head_size"]]
            assert row["exp_sums_shape"] == [ns, nh, parts] and row["exp_sums_dtype"] == "torch.float32"


@pytest.mark.parametrize("case", [0, 1, 2])
def test_fp8_marlin_gemm_oracle_vs_reference_recipe(case):
    """tests/golden/fp8_marlin_*.npz = what the reference's test_fp8_marlin_gemm (tests/kernels/test_marlin_gemm.py:238-304)
    drew and expected when it was run in place (tools/make_golden_fp8_marlin.py); its checker is
    compute_max_diff = mean|out - ref| / mean|ref| < 0.04 (marlin_utils.py compute_max_diff)"""
    g = np.load(os.path.join(GOLD, f"fp8_marlin_{case}.npz"))
    dt = getattr(torch, str(g["dtype"]))
    m, n, k = int(g["m"]), int(g["n"]), int(g["k"])
    a = helpers.from_np(g["a"], dt)
    out = oracle.fp8_marlin_gemm(a, torch.from_numpy(g["marlin_q"].copy()), helpers.from_np(g["marlin_s"], dt), m, n, k)
    ref = helpers.from_np(g["output_ref"], dt)
    diff = ((out.float() - ref.float()).abs().mean() / ref.float().abs().mean()).item()
    assert diff < 0.04
    assert abs(diff - float(g["reference_max_diff_of_oracle"])) < 1e-3
