"""GPU: checkpoint formats -> kernel layouts.  For every quantised checkpoint format
the reference reads for Llama (vllm/model_executor/model_loader/weight_utils.py, models/llama.py:391-515, and
the methods' create_weights / process_weights_after_loading), a synthetic HF-layout checkpoint of the tiny
Llama is written to disk with the format's own tensor names, shapes and config, loaded with
DecodeRunner.from_pretrained, and its prompt logits are compared with a plain fp32 model (tests/ref_llama.py,
itself pinned to the reference's LlamaForCausalLM by tests/test_oracle_golden.py) run on the DEQUANTISED
weights of the same checkpoint:
  * AutoGPTQ (quantize_config.json; symmetric -> gptq_marlin, asymmetric with qzeros -> gptq),
  * AutoAWQ (qweight [K, N/8] in AWQ nibble order, qzeros, scales; fp16),
  * compressed-tensors int-quantized W8A8 (per-channel int8 weights, dynamic per-token activations),
  * compressed-tensors pack-quantized W4A16 (weight_packed [N, K/8], weight_scale [N, K/g], weight_shape),
  * FP8 (serialized e4m3 weights, per-shard weight_scale, static input_scale, per-layer kv_scale),
  * tied embeddings (no lm_head.weight in the checkpoint).
Tolerances are stated per case; weight-only formats share the 3e-2 of the other end-to-end tests."""
import json
import os

import pytest
import torch

import helpers
from oracle import ref_math
from ref_llama import RefLlama

pytestmark = pytest.mark.gpu
A = helpers.TINY_LLAMA


def write_checkpoint(path, tensors, torch_dtype, quantization_config=None, quantize_config=None, **cfg_extra):
    from safetensors.torch import save_file
    names = list(tensors)
    half = len(names) // 2
    save_file({n: tensors[n].contiguous() for n in names[:half]}, os.path.join(path, "model-00001-of-00002.safetensors"))
    save_file({n: tensors[n].contiguous() for n in names[half:]}, os.path.join(path, "model-00002-of-00002.safetensors"))
    cfg = dict(architectures=["LlamaForCausalLM"], hidden_act="silu", torch_dtype=torch_dtype, **A, **cfg_extra)
    if quantization_config is not None:
        cfg["quantization_config"] = quantization_config
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f)
    if quantize_config is not None:
        with open(os.path.join(path, "quantize_config.json"), "w") as f:
            json.dump(quantize_config, f)


def prompt_logits(runner, prompts):
    """[B, V] fp32 logits of the last prompt token through the HIP path (writes the KV cache)"""
    from test_gpu_model_golden import _prefill_metadata
    dev = runner.device
    batch, plen = prompts.shape
    runner.setup_batch(batch, plen, 8)
    md = _prefill_metadata(runner.block_tables, batch, plen, runner.cache_config.block_size, dev)
    with torch.inference_mode():
        hidden = runner.model(prompts.reshape(-1).to(dev), torch.arange(plen, device=dev).repeat(batch),
                              runner.kv_caches, md)
        return runner.model.compute_logits(hidden.view(batch, plen, -1)[:, -1]).float().cpu()[:, :A["vocab_size"]]


def check(runner, dense, tol, what):
    import types
    arch = types.SimpleNamespace(**A, head_dim=A["hidden_size"] // A["num_attention_heads"])
    ref = RefLlama(arch, dense)
    g = torch.Generator().manual_seed(5)
    prompts = torch.randint(0, A["vocab_size"], (3, 19), generator=g)
    got = prompt_logits(runner, prompts)
    want = torch.stack([ref.forward(prompts[b])[-1] for b in range(prompts.shape[0])])
    rel = ((got - want).abs().mean() / want.abs().mean()).item()
    assert rel < tol, f"{what}: prompt logits rel err {rel:.3e} >= {tol}"
    # and a few decode steps run (KV cache of the prompt, paged attention, the method's decode-size kernels)
    runner.input_ids.copy_(got.argmax(-1).to(runner.device))
    for _ in range(3):
        assert runner.decode_step().shape == (3, )
    return rel


def load(tmp_path, device):
    from neural_magic_vllm_amd.worker import decode_runner as dr
    return dr.DecodeRunner.from_pretrained(str(tmp_path), device, cache_config=dr.CacheConfig(16, "auto"))


def linears(ckpt):
    return [n for n in ckpt if helpers.is_quantised_linear(n)]


def test_gptq_asymmetric_checkpoint(gpu_device, tmp_path):
    """AutoGPTQ, sym=False: qweight [K/8, N], qzeros [K/g, N/8] holding z - 1, scales, g_idx -> GPTQLinearMethod"""
    dt = torch.float16
    ckpt = helpers.tiny_llama_checkpoint(3, dt)
    tensors, dense = {}, {}
    for name, w in ckpt.items():
        if not helpers.is_quantised_linear(name):
            tensors[name] = dense[name] = w
            continue
        w_kn = w.t().contiguous()
        k, n = w_kn.shape
        q, z, s = ref_math.quantize_asym(w_kn, 4, 128)
        base = name[:-len(".weight")]
        tensors[base + ".qweight"] = ref_math.gptq_pack(q, 4, k, n)
        tensors[base + ".qzeros"] = ref_math.pack_cols(z - 1, 4)            # GPTQ stores zero - 1
        tensors[base + ".scales"] = s
        tensors[base + ".g_idx"] = torch.arange(k, dtype=torch.int32) // 128
        dense[name] = ref_math.awq_reference_weight(q, z, s, 128, dt).t().contiguous()
    write_checkpoint(str(tmp_path), tensors, "float16",
                     quantize_config=dict(bits=4, group_size=128, desc_act=False, sym=False))
    runner = load(tmp_path, gpu_device)
    from neural_magic_vllm_amd.model_executor.layers.quantization.gptq import GPTQConfig
    assert isinstance(runner.model.model.layers[0].mlp.down_proj.quant_method.quant_config, GPTQConfig)
    check(runner, dense, 3e-2, "gptq asym")


def test_awq_checkpoint(gpu_device, tmp_path):
    dt = torch.float16
    ckpt = helpers.tiny_llama_checkpoint(4, dt)
    tensors, dense = {}, {}
    for name, w in ckpt.items():
        if not helpers.is_quantised_linear(name):
            tensors[name] = dense[name] = w
            continue
        w_kn = w.t().contiguous()
        q, z, s = ref_math.quantize_asym(w_kn, 4, 128)
        base = name[:-len(".weight")]
        tensors[base + ".qweight"] = ref_math.pack_cols(q, 4, ref_math.AWQ_NIBBLE_OF_COLUMN)   # [K, N/8]
        tensors[base + ".qzeros"] = ref_math.pack_cols(z, 4, ref_math.AWQ_NIBBLE_OF_COLUMN)    # [K/g, N/8]
        tensors[base + ".scales"] = s
        dense[name] = ref_math.awq_reference_weight(q, z, s, 128, dt).t().contiguous()
    write_checkpoint(str(tmp_path), tensors, "float16",
                     quantize_config=dict(quant_method="awq", w_bit=4, q_group_size=128, zero_point=True, version="gemm"))
    runner = load(tmp_path, gpu_device)
    from neural_magic_vllm_amd.model_executor.layers.quantization.awq import AWQConfig
    assert isinstance(runner.model.model.layers[0].self_attn.qkv_proj.quant_method.quant_config, AWQConfig)
    check(runner, dense, 3e-2, "awq")


def _ct_config(weights, acts, fmt):
    return dict(quant_method="compressed-tensors", format=fmt, ignore=["lm_head"],
                config_groups={"group_0": {"targets": ["Linear"], "weights": weights, "input_activations": acts}})


def test_compressed_tensors_w8a8_checkpoint(gpu_device, tmp_path):
    dt = torch.bfloat16
    ckpt = helpers.tiny_llama_checkpoint(5, dt)
    tensors, dense = {}, {}
    for name, w in ckpt.items():
        if not helpers.is_quantised_linear(name):
            tensors[name] = dense[name] = w
            continue
        ws = (w.float().abs().amax(dim=1, keepdim=True).clamp_min(1e-8) / 127.0)
        q = torch.clamp(torch.round(w.float() / ws), -127, 127).to(torch.int8)
        base = name[:-len(".weight")]
        tensors[base + ".weight"] = q
        tensors[base + ".weight_scale"] = ws.to(torch.float32)
        dense[base + ".weight"] = q                     # ref_llama's W8A8 branch: int8 x per-channel scale
        dense[base + ".weight_scale"] = ws.to(torch.float32)
    write_checkpoint(str(tmp_path), tensors, "bfloat16", quantization_config=_ct_config(
        dict(num_bits=8, type="int", symmetric=True, strategy="channel", dynamic=False),
        dict(num_bits=8, type="int", symmetric=True, strategy="token", dynamic=True), "int-quantized"))
    runner = load(tmp_path, gpu_device)
    import types
    arch = types.SimpleNamespace(**A, head_dim=A["hidden_size"] // A["num_attention_heads"])
    ref = RefLlama(arch, dense, act_int8=True)     # the reference applies the same per-token int8 rounding
    g = torch.Generator().manual_seed(5)
    prompts = torch.randint(0, A["vocab_size"], (3, 19), generator=g)
    got = prompt_logits(runner, prompts)
    want = torch.stack([ref.forward(prompts[b])[-1] for b in range(3)])
    rel = ((got - want).abs().mean() / want.abs().mean()).item()
    # 8-bit activations on top of 8-bit weights; ref_llama rounds its (fp32) activations per token the same way
    # but at different points of the bf16 pipeline: 6e-2 stated for this format
    assert rel < 6e-2, rel


def test_compressed_tensors_w4a16_checkpoint(gpu_device, tmp_path):
    """pack-quantized: weight_packed int32 [N, K/8] (8 consecutive-K codes of one output row per word, offset
    binary), weight_scale [N, K/g], weight_shape [2] (compressed_tensors_wNa16.py:57-105)"""
    dt = torch.bfloat16
    ckpt = helpers.tiny_llama_checkpoint(6, dt)
    tensors, dense = {}, {}
    for name, w in ckpt.items():
        if not helpers.is_quantised_linear(name):
            tensors[name] = dense[name] = w
            continue
        w_kn = w.t().contiguous()
        k, n = w_kn.shape
        w_ref, q, s, _, _ = ref_math.quantize_weights(w_kn, 4, 128, False)
        base = name[:-len(".weight")]
        tensors[base + ".weight_packed"] = ref_math.pack_cols(q.t().contiguous(), 4)      # [N, K/8]
        tensors[base + ".weight_scale"] = s.t().contiguous().to(dt)                       # [N, K/g]
        tensors[base + ".weight_shape"] = torch.tensor([n, k], dtype=torch.int64)
        dense[name] = w_ref.t().contiguous().to(dt)
    write_checkpoint(str(tmp_path), tensors, "bfloat16", quantization_config=_ct_config(
        dict(num_bits=4, type="int", symmetric=True, strategy="group", group_size=128, dynamic=False), None,
        "pack-quantized"))
    runner = load(tmp_path, gpu_device)
    check(runner, dense, 3e-2, "compressed-tensors w4a16")


def test_fp8_checkpoint_with_kv_scale(gpu_device, tmp_path):
    """serialized FP8: e4m3 weights [N, K], one weight_scale / input_scale per checkpoint shard (q, k, v and gate,
    up arrive separately: fp8.py:249-313 requantises them to one scale), `self_attn.kv_scale` per layer
    (llama.py:470-481 -> Attention._kv_scale with an fp8 KV cache)"""
    from neural_magic_vllm_amd.worker import decode_runner as dr
    dt = torch.bfloat16
    ckpt = helpers.tiny_llama_checkpoint(7, dt)
    tensors, dense = {}, {}
    for name, w in ckpt.items():
        if not helpers.is_quantised_linear(name):
            tensors[name] = dense[name] = w
            continue
        sc = (w.float().abs().max() / 448.0).clamp_min(1e-8)
        q = (w.float() / sc).clamp(-448, 448).to(torch.float8_e4m3fn)
        base = name[:-len(".weight")]
        tensors[base + ".weight"] = q
        tensors[base + ".weight_scale"] = sc.reshape(()).to(torch.float32)
        tensors[base + ".input_scale"] = torch.tensor(0.05, dtype=torch.float32)
        dense[name] = (q.float() * sc).to(dt)
    for i in range(A["num_hidden_layers"]):
        tensors[f"model.layers.{i}.self_attn.kv_scale"] = torch.tensor(0.02 * (i + 1), dtype=torch.float32)
    write_checkpoint(str(tmp_path), tensors, "bfloat16",
                     quantization_config=dict(quant_method="fp8", activation_scheme="static"))
    runner = dr.DecodeRunner.from_pretrained(str(tmp_path), gpu_device, cache_config=dr.CacheConfig(16, "fp8"))
    for i, layer in enumerate(runner.model.model.layers):
        # the parameter is created in the model dtype, as in the reference (fp8.py:575): bf16 precision
        assert abs(layer.self_attn.attn._kv_scale / (0.02 * (i + 1)) - 1) < 2**-8, "kv_scale not loaded"
    # activations are rounded to e4m3 with a fixed scale (3 mantissa bits): the fp32 model on the dequantised
    # weights is only a coarse yardstick for this format
    check(runner, dense, 1.5e-1, "fp8 static + fp8 kv")


def test_tied_embedding_checkpoint(gpu_device, tmp_path):
    dt = torch.bfloat16
    ckpt = helpers.tiny_llama_checkpoint(8, dt)
    tensors = {k: v for k, v in ckpt.items() if k != "lm_head.weight"}
    write_checkpoint(str(tmp_path), tensors, "bfloat16", tie_word_embeddings=True)
    runner = load(tmp_path, gpu_device)
    assert runner.model.lm_head.weight.data_ptr() == runner.model.model.embed_tokens.weight.data_ptr()
    dense = dict(tensors)
    dense["lm_head.weight"] = tensors["model.embed_tokens.weight"]
    check(runner, dense, 3e-2, "tied embeddings")
    # the same checkpoint without the tie flag leaves lm_head unwritten: refused, not served with garbage
    (tmp_path / "config.json").write_text(json.dumps(dict(architectures=["LlamaForCausalLM"], hidden_act="silu",
                                                          torch_dtype="bfloat16", **A)))
    with pytest.raises(ValueError, match="unwritten"):
        load(tmp_path, gpu_device)


def test_unknown_tensor_and_dtype_are_refused(gpu_device, tmp_path):
    dt = torch.bfloat16
    ckpt = helpers.tiny_llama_checkpoint(9, dt)
    write_checkpoint(str(tmp_path), dict(ckpt, **{"model.layers.0.mlp.gate_proj.nonsense": torch.zeros(1)}), "bfloat16")
    with pytest.raises(ValueError, match="no parameter"):
        load(tmp_path, gpu_device)
    write_checkpoint(str(tmp_path), ckpt, "float64")
    with pytest.raises(ValueError, match="torch_dtype"):
        load(tmp_path, gpu_device)
