"""CPU: the LinearMethods create the same parameters (names, shapes, dtypes, sharding attributes)
as the reference's, captured from the reference classes into tests/golden/linear_method_params.json
by tools/make_golden.py (Llama-3-8B qkv projection: K 4096, partitions [4096, 1024, 1024])."""
import json
import os

import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "linear_method_params.json")
KEEP = ("input_dim", "output_dim", "packed_dim", "pack_factor", "marlin_tile_size",
        "needs_scalar_to_array")
# the tables tools/make_golden_w8a8.py adds keep a few more attributes
KEEP_W8A8 = KEEP + ("logical_widths", "ignore_warning", "shard_splitter", "use_bits_and_bytes")


def _ct(weights, acts):
    return {"config_groups": {"group_0": {"targets": ["Linear"], "weights": weights, "input_activations": acts}},
            "format": "int-quantized" if acts is not None else "pack-quantized", "ignore": ["lm_head"]}


_W8 = dict(num_bits=8, type="int", symmetric=True, strategy="tensor", dynamic=False)
_W8C = dict(num_bits=8, type="int", symmetric=True, strategy="channel", dynamic=False)
CFGS = {
    "gptq_marlin": ("gptq_marlin", dict(bits=4, group_size=128, desc_act=False, sym=True), torch.bfloat16),
    "gptq_marlin_act_order": ("gptq_marlin", dict(bits=4, group_size=128, desc_act=True, sym=True), torch.bfloat16),
    "gptq": ("gptq", dict(bits=4, group_size=128, desc_act=False), torch.float16),
    "awq": ("awq", dict(w_bit=4, q_group_size=128, zero_point=True), torch.bfloat16),
    "fp8_static": ("fp8", dict(quant_method="fp8", activation_scheme="static"), torch.bfloat16),
    "fp8_dynamic": ("fp8", dict(quant_method="fp8", activation_scheme="dynamic"), torch.bfloat16),
    "marlin": ("marlin", dict(group_size=128), torch.float16),
    "ct_w8a8_static": ("compressed-tensors",
                       _ct(_W8, dict(num_bits=8, type="int", symmetric=True, strategy="tensor", dynamic=False)),
                       torch.bfloat16),
    "ct_w8a8_dynamic_token": ("compressed-tensors",
                              _ct(_W8C, dict(num_bits=8, type="int", symmetric=True, strategy="token", dynamic=True)),
                              torch.bfloat16),
    "ct_w4a16_g128": ("compressed-tensors",
                      _ct(dict(num_bits=4, type="int", symmetric=True, strategy="group", group_size=128,
                               dynamic=False), None), torch.bfloat16),
    "ct_w8a16_channel": ("compressed-tensors",
                         _ct(dict(num_bits=8, type="int", symmetric=True, strategy="channel", dynamic=False), None),
                         torch.bfloat16),
}
OLD_TABLES = ("gptq_marlin", "gptq_marlin_act_order", "gptq", "awq")


@pytest.mark.parametrize("name", sorted(CFGS))
def test_parameter_tables_match_reference(name):
    from neural_magic_vllm_amd.model_executor.layers.linear import ColumnParallelLinear
    from neural_magic_vllm_amd.model_executor.layers.quantization import get_quantization_config
    gold = json.load(open(GOLD))[name]
    method, cfg, dtype = CFGS[name]
    qc = get_quantization_config(method).from_config(cfg)

    class Dummy(torch.nn.Module):
        pass

    layer = Dummy()
    lm = qc.get_quant_method(ColumnParallelLinear(64, 64, bias=False, params_dtype=dtype,
                                                  quant_config=None))
    assert lm is not None
    lm.create_weights(layer, 4096, [4096, 1024, 1024], 4096, 6144, dtype, weight_loader=None)
    got = {}
    for pname, prm in layer.named_parameters():
        attrs = {}
        for k in (KEEP if name in OLD_TABLES else KEEP_W8A8):
            if hasattr(prm, k):
                v = getattr(prm, k)
                attrs[k] = str(v) if k == "pack_factor" else (v if isinstance(v, (int, bool, list, type(None)))
                                                              else type(v).__name__)
        got[pname] = dict(shape=list(prm.shape), dtype=str(prm.dtype), device=prm.device.type, attrs=attrs)
    assert got == gold, {k: (got.get(k), gold.get(k)) for k in set(got) | set(gold) if got.get(k) != gold.get(k)}


def test_every_reference_table_is_checked():
    assert sorted(json.load(open(GOLD))) == sorted(CFGS)


def test_pack_fp8_to_int32_matches_reference():
    """fp8.py's host-side packer (marlin_utils.py:227-247) against the reference's output"""
    import numpy as np
    from neural_magic_vllm_amd.model_executor.layers.quantization.fp8 import pack_fp8_to_int32
    g = np.load(os.path.join(os.path.dirname(GOLD), "fp8_quant.npz"))
    w = torch.from_numpy(g["pack_in"].copy()).view(torch.float8_e4m3fn)
    assert torch.equal(pack_fp8_to_int32(w), torch.from_numpy(g["pack_out"]))


def test_registry_and_configs():
    from neural_magic_vllm_amd.model_executor.layers.quantization import (QUANTIZATION_METHODS,
                                                                          get_quantization_config)
    assert list(QUANTIZATION_METHODS) == ["awq", "fp8", "marlin", "gptq_marlin", "gptq",
                                          "compressed-tensors"]
    with pytest.raises(ValueError):
        get_quantization_config("aqlm")
    gm = get_quantization_config("gptq_marlin")
    assert gm.is_marlin_compatible(dict(bits=4, group_size=128, sym=True, desc_act=False))
    assert not gm.is_marlin_compatible(dict(bits=3, group_size=128, sym=True, desc_act=False))
    assert gm.override_quantization_method(dict(bits=4, group_size=128, sym=True, desc_act=True), None) == "gptq_marlin"
    ct = get_quantization_config("compressed-tensors").from_config({
        "format": "int-quantized",
        "config_groups": {"g0": {"targets": ["Linear"],
                                 "weights": {"num_bits": 8, "type": "int", "symmetric": True, "strategy": "channel"},
                                 "input_activations": {"num_bits": 8, "type": "int", "dynamic": True, "strategy": "token"}}}})
    sch = ct.get_scheme(None)
    assert type(sch).__name__ == "CompressedTensorsW8A8" and not sch.is_static_input_scheme


def test_paged_attention_v1_v2_heuristic():
    """truth table of the reference's choice (paged_attn.py:112-121)"""
    from neural_magic_vllm_amd.attention.ops.paged_attn import PagedAttention
    cases = {(512, 1, 32): True, (513, 1, 32): False, (513, 16, 32): False, (513, 17, 32): True,
             (8192, 64, 32): True, (8193, 64, 32): False, (4096, 8, 64): False, (100, 1, 1): True}
    for (msl, ns, nh), exp in cases.items():
        assert PagedAttention.use_v1(msl, ns, nh) == exp, (msl, ns, nh)
    assert PagedAttention.get_kv_cache_shape(10, 16, 8, 128) == (2, 10, 16 * 8 * 128)


def test_fused_attention_launch_form_and_native_copy_ranges(monkeypatch):
    """this repo's own choices beside the reference's rules: the fused rope + cache + attention launch stays
    unpartitioned up to 896 tokens (measured on MI355X) and follows the reference's rule beyond; the
    MFMA-native copy of the 4-bit weights serves every decode-sized call, up to 64 rows"""
    from types import SimpleNamespace

    from neural_magic_vllm_amd.attention.ops.paged_attn import PagedAttention
    from neural_magic_vllm_amd.model_executor.layers.quantization.gptq_marlin import GPTQMarlinLinearMethod as LM
    for (msl, ns, nh), exp in {(512, 1, 32): True, (530, 1, 32): True, (896, 16, 32): True, (897, 16, 32): False,
                               (897, 17, 32): True, (4096, 1, 32): False, (8193, 64, 32): False}.items():
        assert PagedAttention.use_v1_fused(msl, ns, nh) == exp, (msl, ns, nh)
        assert PagedAttention.use_v1_fused(msl, ns, nh) or not PagedAttention.use_v1(msl, ns, nh)   # never stricter
    # a layer that keeps BOTH tensors (NMV_W4_KEEP_MARLIN=1) sends a prompt-sized call to the native tensor where its
    # kernels were measured ahead (512 rows and 64 large tiles or more: Llama-3-8B gate_up from 512 rows, the narrow
    # projections from 1024 / 2048); a layer that dropped the Marlin tensor (the default) sends everything there
    def layer(n, k, **kw):
        return SimpleNamespace(output_size_per_partition=n, input_size_per_partition=k, **kw)
    gate_up, o_proj = layer(28672, 4096, qweight_native=object()), layer(4096, 4096, qweight_native=object())
    assert [m for m in (1, 16, 17, 32, 33, 64, 65, 255, 256, 512, 4096) if LM._native(gate_up, m)] == \
        [1, 16, 17, 32, 33, 64, 512, 4096]
    assert [m for m in (1, 16, 17, 32, 33, 64, 65, 512, 2048, 4096) if LM._native(o_proj, m)] == [1, 16, 17, 32, 33, 64, 2048, 4096]
    only = layer(4096, 4096, qweight_native=object(), marlin_dropped=True)
    assert all(LM._native(only, m) for m in (1, 64, 65, 100, 512, 100000))
    monkeypatch.setenv("NMV_W4P", "0")
    assert not LM._native(gate_up, 512)
    assert not any(LM._native(layer(28672, 4096), m) for m in (1, 16, 64, 512))


def test_deferred_reduce_admission_asks_the_method_with_the_layers_group_count(monkeypatch):
    """linear.py's admission check for the deferred split-K forms goes through the LinearMethod's can_defer(layer, rows)
    hook -- the planner apply_partial will run, with the layer's own scale-group count (a channelwise layer has ONE) --
    and methods without the hook are asked through the splits query with that count, never with the group-128 default"""
    from types import SimpleNamespace

    import torch

    from neural_magic_vllm_amd.model_executor.layers import linear
    seen = []
    monkeypatch.setattr(linear.ops, "gptq_marlin_gemm_partial_splits",
                        lambda rows, n, k, groups=None: seen.append((rows, n, k, groups)) or 1)
    channelwise = SimpleNamespace(scales=torch.zeros(1, 512))
    grouped = SimpleNamespace(scales=torch.zeros(8, 512))
    no_hook = SimpleNamespace()
    assert linear._method_can_defer(no_hook, channelwise, 5, 512, 1024)
    assert linear._method_can_defer(no_hook, grouped, 5, 512, 1024)
    assert seen == [(5, 512, 1024, 1), (5, 512, 1024, 8)]
    hooked = SimpleNamespace(can_defer=lambda layer, rows: rows <= 4)
    assert linear._method_can_defer(hooked, grouped, 4, 512, 1024) and not linear._method_can_defer(hooked, grouped, 5, 512, 1024)
    assert len(seen) == 2   # the hook decides alone


def test_custom_allreduce_mesh_check_uses_physical_device_ids(monkeypatch):
    """CustomAllreduce.full_nvlink (round-3 ADVICE): ranks exchange PHYSICAL device ids (through HIP_VISIBLE_DEVICES /
    CUDA_VISIBLE_DEVICES, as the reference's _can_p2p does), and a peer device this process cannot see makes the answer
    'unknown' (None), never 'full mesh' -- with one visible device per rank every local ordinal is 0"""
    import torch

    from neural_magic_vllm_amd.distributed.device_communicators import custom_all_reduce as car
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "5,2")
    assert car._visible_physical_ids() == [5, 2]
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.setenv("CUDA_VISIBLE_DEVICES", "3")
    assert car._visible_physical_ids() == [3]
    monkeypatch.setenv("CUDA_VISIBLE_DEVICES", "GPU-deadbeef")
    assert car._visible_physical_ids() == []
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES")
    assert car._visible_physical_ids() == [0, 1]
    asked = []
    monkeypatch.setattr(torch.cuda, "can_device_access_peer", lambda a, b: asked.append((a, b)) or (a, b) != (0, 1))
    assert car._mesh_is_full([5, 2], [5, 2]) is False and (0, 1) in asked     # asked with LOCAL ordinals of physical 5, 2
    assert car._mesh_is_full([2, 2, 2, 2], [5, 2]) is True                    # ranks time-sharing one device (tests)
    assert car._mesh_is_full([0, 1, 2, 3], [1]) is None                       # one visible device per rank: unknown


def test_checkpoint_iterator_and_quant_config(tmp_path):
    """model_loader: every tensor of every safetensors shard comes back under its name; the quantisation
    config is config.json's `quantization_config` when present, else quantize_config.json
    (weight_utils.py get_quant_config)"""
    import json

    import torch
    from safetensors.torch import save_file
    from neural_magic_vllm_amd.model_executor import model_loader as ml
    a = {"model.layers.0.self_attn.q_proj.qweight": torch.arange(12, dtype=torch.int32).view(3, 4),
         "model.norm.weight": torch.ones(8, dtype=torch.bfloat16)}
    b = {"lm_head.weight": torch.randn(4, 8).to(torch.bfloat16)}
    save_file(a, str(tmp_path / "model-00001-of-00002.safetensors"))
    save_file(b, str(tmp_path / "model-00002-of-00002.safetensors"))
    got = dict(ml.safetensors_weights_iterator(str(tmp_path)))
    assert set(got) == set(a) | set(b)
    for k, v in {**a, **b}.items():
        assert torch.equal(got[k], v) and got[k].dtype == v.dtype
    (tmp_path / "config.json").write_text(json.dumps(dict(hidden_size=8)))
    cfg = ml.read_hf_config(str(tmp_path))
    assert ml.read_quant_config(str(tmp_path), cfg) is None
    (tmp_path / "quantize_config.json").write_text(json.dumps(dict(bits=4, group_size=128, desc_act=False, sym=True)))
    assert ml.read_quant_config(str(tmp_path), cfg)["bits"] == 4
    cfg["quantization_config"] = dict(quant_method="fp8", activation_scheme="dynamic")
    assert ml.read_quant_config(str(tmp_path), cfg)["quant_method"] == "fp8"      # config.json wins
    with __import__("pytest").raises(FileNotFoundError):
        list(ml.safetensors_weights_iterator(str(tmp_path / "nowhere")))
