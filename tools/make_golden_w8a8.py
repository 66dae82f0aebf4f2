"""Generate the fixtures that pin the W8A8 leg, the small quantisers, the v1/v2 choice and the remaining
LinearMethod parameter tables to the REFERENCE's own code.  Runs only in the build container.

How the reference is used (its CUDA kernels cannot run here; its Python can):
  * tests/kernels/test_cutlass.py: the functions `to_int8`, `to_fp8`, `baseline_scaled_mm`,
    `cutlass_int8_gemm_helper`, `cutlass_fp8_gemm_helper` are compiled from the reference's file IN PLACE
    (ast -> exec; nothing is copied into this repo) and the helpers are run on CPU with
    `ops.cutlass_scaled_mm` bound to this repo's oracle: the reference's own assert
    (`torch.allclose(out, baseline, ...)`, :79 / :110) then judges the oracle, and the inputs the helper
    drew plus `baseline_scaled_mm`'s output go into tests/golden/scaled_mm_*.npz.
  * tests/kernels/test_int8_quant.py: `test_dynamic_scaled_int8_quant` / `test_static_scaled_int8_quant`
    are run the same way (device constant rewritten to "cpu", `torch.ops._C.*_scaled_int8_quant` bound to
    the oracle); the expected tensors the test computes (`torch_out`, `scales`, `out1`) are read out of the
    test's frame when it calls the op, so the fixture holds the reference test's own expectation.
  * vllm/model_executor/layers/quantization/fp8.py `per_tensor_quantize`, utils/marlin_utils.py
    `pack_fp8_to_int32`: imported and called.
  * vllm/attention/ops/paged_attn.py `PagedAttention.forward_decode`: called on a grid of
    (num_seqs, num_heads, max_seq_len) with `ops.paged_attention_v1/v2` replaced by recorders -> the v1/v2
    truth table and the tmp-buffer shapes (tests/golden/pa_heuristic.json).
  * the fp8 / marlin / compressed-tensors LinearMethods: parameter tables, merged into
    tests/golden/linear_method_params.json next to the four that tools/make_golden.py writes.

usage:  python tools/make_golden_w8a8.py
"""
import ast
import json
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLD = os.path.join(ROOT, "tests", "golden")

import helpers  # noqa: E402
import oracle  # noqa: E402
from oracle import build_ref  # noqa: E402

REF = build_ref.REF_ROOT


def save(name, **arrays):
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def functions_of(path, names, rewrite_cuda=False):
    """compile the named top-level functions of a reference file in place and return them (decorators
    dropped; optionally every string constant "cuda" turned into "cpu")"""
    tree = ast.parse(open(path).read(), filename=path)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(n.name for n in keep) == sorted(names), (path, [n.name for n in keep])
    for n in keep:
        n.decorator_list = []

    class ToCpu(ast.NodeTransformer):
        def visit_Constant(self, node):
            return ast.copy_location(ast.Constant("cpu"), node) if node.value == "cuda" else node

    mod = ast.Module(body=keep, type_ignores=[])
    if rewrite_cuda:
        mod = ToCpu().visit(mod)
    ast.fix_missing_locations(mod)
    return compile(mod, path, "exec")


# ------------------------------------------------------------------------------------------------
def gen_scaled_mm():
    from typing import Optional, Type
    recorded = {}

    class Ops:
        @staticmethod
        def cutlass_scaled_mm(a, b, scale_a, scale_b, out_dtype, bias=None):
            recorded.update(a=a, b=b, scale_a=scale_a, scale_b=scale_b, out_dtype=out_dtype, bias=bias)
            return oracle.scaled_mm(a, b, scale_a, scale_b, out_dtype, bias)

    ns = dict(torch=torch, Optional=Optional, Type=Type, ops=Ops)
    exec(functions_of(os.path.join(REF, "tests/kernels/test_cutlass.py"),
                      ["to_fp8", "to_int8", "baseline_scaled_mm", "cutlass_fp8_gemm_helper",
                       "cutlass_int8_gemm_helper"]), ns)
    # (kind, m, n, k, per_token, per_channel, bias, out dtype): the reference grid's corners (test_cutlass.py
    # :113-140), incl. per-token x per-channel x bias, M = 1 and a K that is not a multiple of 128
    cases = [("int8", 33, 256, 128, True, True, True, torch.bfloat16),
             ("int8", 1, 256, 496, False, False, False, torch.bfloat16),
             ("int8", 222, 256, 496, True, False, True, torch.float16),
             ("int8", 100, 1024, 128, False, True, False, torch.bfloat16),
             ("int8", 64, 256, 1024, True, True, False, torch.float16),
             ("fp8", 33, 256, 128, True, True, True, torch.bfloat16),
             ("fp8", 1, 256, 496, False, False, False, torch.bfloat16),
             ("fp8", 100, 256, 496, True, False, True, torch.float16),
             ("fp8", 64, 1024, 128, False, True, False, torch.bfloat16)]
    for i, (kind, m, n, k, pt, pc, ub, dt) in enumerate(cases):
        torch.manual_seed(100 + i)
        recorded.clear()
        ns[f"cutlass_{kind}_gemm_helper"](m, n, k, pt, pc, ub, dt, "cpu")   # asserts oracle vs baseline itself
        r = dict(recorded)
        base = ns["baseline_scaled_mm"](r["a"], r["b"], r["scale_a"], r["scale_b"], r["out_dtype"], r["bias"])
        a_u8 = r["a"].view(torch.uint8) if kind == "fp8" else r["a"].view(torch.uint8)
        bt_u8 = r["b"].t().contiguous().view(torch.uint8)      # [N, K] row-major = b column-major
        save(f"scaled_mm_{kind}_{i}", kind=kind, m=m, n=n, k=k, per_token=pt, per_channel=pc, use_bias=ub,
             out_dtype=str(dt).replace("torch.", ""), a_bytes=a_u8.numpy(), b_t_bytes=bt_u8.numpy(),
             scale_a=r["scale_a"].numpy(), scale_b=r["scale_b"].numpy(),
             bias=helpers.to_np(r["bias"]) if ub else np.zeros(0, dtype=np.int16),
             baseline=helpers.to_np(base))


# ------------------------------------------------------------------------------------------------
def gen_int8_quant():
    """run the reference's two int8-quant tests on CPU against the oracle and keep what they expected"""
    import inspect
    got = {}

    def frame_of(name):
        f = inspect.currentframe()
        while f is not None and f.f_code.co_name != name:
            f = f.f_back
        assert f is not None, name
        return f.f_locals

    def dynamic(out, x, scale):
        loc = frame_of("test_dynamic_scaled_int8_quant")
        got.update(x=x.clone(), expect_q=loc["torch_out"].clone(), expect_s=loc["scales"].clone())
        q, s = oracle.scaled_int8_quant(x)
        out.copy_(q)
        scale.copy_(s)

    def static(out, x, scale):
        loc = frame_of("test_static_scaled_int8_quant")
        got.update(x=x.clone(), expect_q=loc["out1"].clone(), scale=float(loc["scale"]))
        q, _ = oracle.scaled_int8_quant(x, scale.clone())
        out.copy_(q)

    lib = torch.library.Library("_C", "DEF")
    lib.define("dynamic_scaled_int8_quant(Tensor! out, Tensor input, Tensor! scale) -> ()")
    lib.define("static_scaled_int8_quant(Tensor! out, Tensor input, Tensor scale) -> ()")
    lib.impl("dynamic_scaled_int8_quant", dynamic, "CPU")
    lib.impl("static_scaled_int8_quant", static, "CPU")
    ns = dict(torch=torch)
    exec(functions_of(os.path.join(REF, "tests/kernels/test_int8_quant.py"),
                      ["test_dynamic_scaled_int8_quant", "test_static_scaled_int8_quant"], rewrite_cuda=True), ns)
    out = {}
    # the reference grid's hidden sizes incl. the odd ones (:12-13); fp32 inputs are not stored (the op's
    # fp32 path is the same arithmetic), token counts kept small
    for j, (nt, hs, dt) in enumerate([(7, 16, torch.bfloat16), (7, 67, torch.float16), (83, 768, torch.bfloat16),
                                      (7, 5137, torch.float16), (1, 8193, torch.bfloat16)]):
        got.clear()
        ns["test_dynamic_scaled_int8_quant"](nt, hs, dt, 0)       # the reference's asserts judge the oracle
        out[f"dyn{j}_x"] = helpers.to_np(got["x"])
        out[f"dyn{j}_q"] = got["expect_q"].numpy()
        out[f"dyn{j}_s"] = got["expect_s"].numpy()
        out[f"dyn{j}_dtype"] = str(dt).replace("torch.", "")
    for j, (nt, hs, dt, sc) in enumerate([(7, 16, torch.bfloat16, 0.1), (83, 67, torch.float16, 0.8),
                                          (7, 2048, torch.bfloat16, 2.1), (1, 8193, torch.float16, 1.2)]):
        got.clear()
        ns["test_static_scaled_int8_quant"](nt, hs, dt, 0, sc)
        out[f"sta{j}_x"] = helpers.to_np(got["x"])
        out[f"sta{j}_q"] = got["expect_q"].numpy()
        out[f"sta{j}_scale"] = np.float32(got["scale"])
        out[f"sta{j}_dtype"] = str(dt).replace("torch.", "")
    save("int8_quant", n_dynamic=5, n_static=4, **out)


# ------------------------------------------------------------------------------------------------
def load_reference_python():
    sys.modules.setdefault("cpuinfo", types.ModuleType("cpuinfo"))  # optional dep, absent here
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import vllm  # noqa: F401


def gen_fp8():
    load_reference_python()
    from vllm.model_executor.layers.quantization.fp8 import per_tensor_quantize
    from vllm.model_executor.layers.quantization.utils.marlin_utils import pack_fp8_to_int32
    g = torch.Generator().manual_seed(7)
    out = {}
    for j, (shape, dt, inv_scale) in enumerate([((5, 64), torch.bfloat16, 0.02), ((33, 257), torch.float16, 0.5),
                                                ((4, 1024), torch.bfloat16, 1e-3)]):
        x = (torch.randn(shape, generator=g) * 3).to(dt)
        x[0, 0], x[0, 1] = 1e4, -1e4                         # saturates at +-448
        q = per_tensor_quantize(x, inv_scale)                  # fp8.py:601-605
        out[f"ptq{j}_x"] = helpers.to_np(x)
        out[f"ptq{j}_inv_scale"] = np.float32(inv_scale)
        out[f"ptq{j}_q"] = q.view(torch.uint8).numpy()
        out[f"ptq{j}_dtype"] = str(dt).replace("torch.", "")
    w = torch.randn((64, 48), generator=g).to(torch.float8_e4m3fn)
    out["pack_in"] = w.view(torch.uint8).numpy()
    out["pack_out"] = pack_fp8_to_int32(w).numpy()             # marlin_utils.py:227-247
    save("fp8_quant", n_ptq=3, **out)


def gen_pa_heuristic():
    load_reference_python()
    import vllm.attention.ops.paged_attn as pa
    calls = []

    def v1(out, q, kc, vc, nkv, scale, bt, sl, bs, msl, *rest):
        calls.append(dict(use_v1=True, partitions=None, tmp_shape=None))

    def v2(out, exp_sums, max_logits, tmp_out, q, kc, vc, nkv, scale, bt, sl, bs, msl, *rest):
        calls.append(dict(use_v1=False, partitions=int(exp_sums.shape[2]), tmp_shape=list(tmp_out.shape),
                          exp_sums_shape=list(exp_sums.shape), exp_sums_dtype=str(exp_sums.dtype),
                          tmp_dtype=str(tmp_out.dtype)))

    table = []
    with mock.patch.object(pa.ops, "paged_attention_v1", v1, create=True), \
            mock.patch.object(pa.ops, "paged_attention_v2", v2, create=True):
        for num_seqs in (1, 2, 8, 16, 17, 64, 256):
            for num_heads in (8, 32, 64):
                for msl in (1, 511, 512, 513, 1024, 4096, 8192, 8193, 16384):
                    hs, bs, nkv = 128, 16, 8
                    q = torch.empty((num_seqs, num_heads, hs), dtype=torch.bfloat16)
                    kc = torch.empty((1, nkv, hs // 8, bs, 8), dtype=torch.bfloat16)
                    vc = torch.empty((1, nkv, hs, bs), dtype=torch.bfloat16)
                    bt = torch.zeros((num_seqs, 1), dtype=torch.int32)
                    sl = torch.ones(num_seqs, dtype=torch.int32)
                    calls.clear()
                    pa.PagedAttention.forward_decode(q, kc, vc, bt, sl, msl, "auto", nkv, hs**-0.5, None, 1.0)
                    assert len(calls) == 1
                    table.append(dict(num_seqs=num_seqs, num_heads=num_heads, max_seq_len=msl, **calls[0]))
    with open(os.path.join(GOLD, "pa_heuristic.json"), "w") as f:
        json.dump(dict(head_size=128, block_size=16, dtype="torch.bfloat16", table=table), f, indent=0)
    print(f"  pa_heuristic.json  {len(table)} rows")


def gen_param_tables():
    """fp8 (serialized checkpoint, static + dynamic activations), legacy marlin, compressed-tensors W8A8
    (static per-tensor, dynamic per-token) and WNA16 (group 128, channelwise): names / shapes / dtypes /
    loader attributes of the parameters the reference's classes create for the Llama-3-8B qkv projection"""
    load_reference_python()
    from vllm.model_executor.layers.quantization import QUANTIZATION_METHODS
    from vllm.platforms import current_platform
    keep = ("input_dim", "output_dim", "packed_dim", "pack_factor", "marlin_tile_size", "needs_scalar_to_array",
            "logical_widths", "ignore_warning", "shard_splitter", "use_bits_and_bytes")

    def ct(weights, acts):
        return {"config_groups": {"group_0": {"targets": ["Linear"], "weights": weights, "input_activations": acts}},
                "format": "int-quantized" if acts is not None else "pack-quantized", "ignore": ["lm_head"]}

    w8 = dict(num_bits=8, type="int", symmetric=True, strategy="tensor", dynamic=False)
    w8c = dict(num_bits=8, type="int", symmetric=True, strategy="channel", dynamic=False)
    cfgs = {
        "fp8_static": ("fp8", dict(quant_method="fp8", activation_scheme="static"), torch.bfloat16),
        "fp8_dynamic": ("fp8", dict(quant_method="fp8", activation_scheme="dynamic"), torch.bfloat16),
        "marlin": ("marlin", dict(group_size=128), torch.float16),
        "ct_w8a8_static": ("compressed-tensors",
                           ct(w8, dict(num_bits=8, type="int", symmetric=True, strategy="tensor", dynamic=False)),
                           torch.bfloat16),
        "ct_w8a8_dynamic_token": ("compressed-tensors",
                                  ct(w8c, dict(num_bits=8, type="int", symmetric=True, strategy="token", dynamic=True)),
                                  torch.bfloat16),
        "ct_w4a16_g128": ("compressed-tensors",
                          ct(dict(num_bits=4, type="int", symmetric=True, strategy="group", group_size=128,
                                  dynamic=False), None), torch.bfloat16),
        "ct_w8a16_channel": ("compressed-tensors",
                             ct(dict(num_bits=8, type="int", symmetric=True, strategy="channel", dynamic=False), None),
                             torch.bfloat16),
    }
    path = os.path.join(GOLD, "linear_method_params.json")
    out = json.load(open(path)) if os.path.exists(path) else {}

    class DummyLinear(torch.nn.Module):   # compressed-tensors matches its "Linear" target on the class name
        pass

    Dummy = DummyLinear
    real_zeros, real_empty = torch.zeros, torch.empty

    def on_cpu(fn):
        def wrapped(*a, **k):
            if str(k.get("device", "")).startswith("cuda"):
                k["device"] = "cpu"
            return fn(*a, **k)
        return wrapped

    # gfx950 reports capability 9.5; the CUDA-only helpers these constructors touch are answered on CPU
    with mock.patch.object(type(current_platform), "get_device_capability", staticmethod(lambda device_id=0: (9, 5))), \
            mock.patch("torch.zeros", on_cpu(real_zeros)), mock.patch("torch.empty", on_cpu(real_empty)), \
            mock.patch("torch.cuda.get_device_capability", lambda *a, **k: (9, 5)), \
            mock.patch("vllm._custom_ops.cutlass_scaled_mm_supports_fp8", lambda cap: True):
        for name, (method, cfg, dtype) in cfgs.items():
            qc = QUANTIZATION_METHODS[method].from_config(cfg)
            import importlib
            mod = importlib.import_module(type(qc).__module__)
            lm_cls = [getattr(mod, n) for n in dir(mod) if n.endswith("LinearMethod") and n != "LinearMethodBase"][0]
            layer = Dummy()
            lm = lm_cls(qc)
            lm.create_weights(layer, 4096, [4096, 1024, 1024], 4096, 6144, dtype, weight_loader=None)
            table = {}
            for pname, prm in layer.named_parameters():
                attrs = {}
                for k in keep:
                    if hasattr(prm, k):
                        v = getattr(prm, k)
                        attrs[k] = str(v) if k == "pack_factor" else (v if isinstance(v, (int, bool, list, type(None)))
                                                                      else type(v).__name__)
                table[pname] = dict(shape=list(prm.shape), dtype=str(prm.dtype), device=prm.device.type, attrs=attrs)
            out[name] = table
            print(f"  {name}: {sorted(table)}")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("  linear_method_params.json", sorted(out))


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    assert build_ref.have_reference(), "/root/reference is not here"
    oracle.build()
    if "--tables-only" not in sys.argv:
        gen_scaled_mm()
        gen_int8_quant()
        gen_fp8()
        gen_pa_heuristic()
    gen_param_tables()
    print("done")
