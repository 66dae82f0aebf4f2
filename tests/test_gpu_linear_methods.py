"""GPU: every quantisation LinearMethod of the plugin surface end to end -- create_weights, the
weight loaders, process_weights_after_loading / lazy repack, apply -- against dense math on the
same synthetic checkpoint tensors."""
import pytest
import torch

from oracle import ref_math

pytestmark = pytest.mark.gpu


def make_layer(method, cfg, k, n_parts, dtype, dev):
    from neural_magic_vllm_amd.model_executor.layers.linear import MergedColumnParallelLinear
    from neural_magic_vllm_amd.model_executor.layers.quantization import get_quantization_config
    qc = get_quantization_config(method).from_config(cfg)
    with torch.device(dev):
        layer = MergedColumnParallelLinear(k, n_parts, bias=False, params_dtype=dtype, quant_config=qc)
    return layer


def finish(layer):
    layer.quant_method.process_weights_after_loading(layer)


@pytest.mark.parametrize("desc_act", [False, True])
@pytest.mark.parametrize("m", [3, 40])
def test_gptq_marlin_method(gpu_device, desc_act, m):
    k, parts, dt = 512, [256, 128], torch.bfloat16
    layer = make_layer("gptq_marlin", dict(bits=4, group_size=128, desc_act=desc_act, sym=True), k, parts, dt, gpu_device)
    g = torch.Generator().manual_seed(0)
    refs = []
    for i, n in enumerate(parts):
        w = torch.randn((k, n), generator=g).to(dt)
        w_ref, q, s, g_idx, _ = ref_math.quantize_weights(w, 4, 128, desc_act, g)
        if not desc_act:
            g_idx = (torch.arange(k) // 128).to(torch.int32)
        layer.qweight.weight_loader(layer.qweight, ref_math.gptq_pack(q, 4, k, n), i)
        layer.scales.weight_loader(layer.scales, s, i)
        if i == 0:
            g0 = g_idx
            layer.g_idx.weight_loader(layer.g_idx, g_idx)
        refs.append((q, s, g_idx))
    # with act-order every fused shard must share one g_idx (as real checkpoints do): rebuild shard 1
    if desc_act:
        w = torch.randn((k, parts[1]), generator=g).to(dt)
        gs = torch.arange(k) // 128
        wq, q1, s1, _, _ = ref_math.quantize_weights(w, 4, 128, False)
        inv = torch.empty(k, dtype=torch.long)
        q1p = q1[torch.argsort(torch.argsort(g0.long(), stable=True), stable=True)] if False else None
        # rows of shard 1 in the same (shuffled) order as shard 0: row j belongs to group g0[j]
        order = torch.argsort(g0.long(), stable=True)
        q1s = torch.empty_like(q1)
        q1s[order] = q1  # sorted position r -> shuffled row order[r]
        layer.qweight.weight_loader(layer.qweight, ref_math.gptq_pack(q1s, 4, k, parts[1]), 1)
        layer.scales.weight_loader(layer.scales, s1, 1)
        refs[1] = (q1s, s1, g0)
    finish(layer)
    x = torch.randn((m, k), generator=g).to(dt)
    out, _ = layer(x.to(gpu_device))
    w_full = torch.cat([((q.float() - 8) * s.float()[gi.long()]).to(dt) for q, s, gi in refs], dim=1)
    ref = x.float() @ w_full.float()
    assert ref_math.compute_max_diff(out.cpu(), ref) < 6e-3
    # second call takes the READY path
    out2, _ = layer(x.to(gpu_device))
    assert torch.equal(out, out2)


@pytest.mark.parametrize("desc_act", [False, True])
def test_gptq_method(gpu_device, desc_act):
    k, parts, dt, m = 512, [256, 128], torch.float16, 9
    layer = make_layer("gptq", dict(bits=4, group_size=128, desc_act=desc_act), k, parts, dt, gpu_device)
    g = torch.Generator().manual_seed(1)
    g_idx = (torch.arange(k) // 128).to(torch.int32)
    if desc_act:
        g_idx = g_idx[torch.randperm(k, generator=g)]
    layer.g_idx.weight_loader(layer.g_idx, g_idx)
    ws = []
    for i, n in enumerate(parts):
        w = torch.randn((k, n), generator=g).to(dt)
        q, z, s = ref_math.quantize_asym(w, 4, 128)
        if desc_act:  # store rows in shuffled order: row j uses group g_idx[j]
            order = torch.argsort(g_idx.long(), stable=True)
            qs = torch.empty_like(q)
            qs[order] = q
            q = qs
        layer.qweight.weight_loader(layer.qweight, ref_math.gptq_pack(q, 4, k, n), i)
        layer.qzeros.weight_loader(layer.qzeros, ref_math.pack_cols(z - 1, 4), i)
        layer.scales.weight_loader(layer.scales, s, i)
        ws.append(ref_math.gptq_reference_weight(q, z, s, g_idx, dt))
    finish(layer)
    # without act-order the layer was repacked to Marlin + zero points at load; with it, exllama path
    assert (getattr(layer, "gptq_marlin_kn", None) is not None) == (not desc_act)
    x = torch.randn((m, k), generator=g).to(dt)
    out, _ = layer(x.to(gpu_device))
    assert ref_math.compute_max_diff(out.cpu(), x.float() @ torch.cat(ws, 1).float()) < 5e-3


@pytest.mark.parametrize("m", [5, 300])
@pytest.mark.parametrize("k", [512, 384])
def test_awq_method(gpu_device, m, k):
    """k = 512: repacked to Marlin at load (zero-point Marlin kernel); k = 384 (not a multiple of
    256): stays on ops.awq_gemm / dequantize + matmul like the reference"""
    parts, dt = [256, 128], torch.float16
    layer = make_layer("awq", dict(w_bit=4, q_group_size=128, zero_point=True), k, parts, dt, gpu_device)
    g = torch.Generator().manual_seed(2)
    ws = []
    for i, n in enumerate(parts):
        w = torch.randn((k, n), generator=g).to(dt)
        q, z, s = ref_math.quantize_asym(w, 4, 128)
        layer.qweight.weight_loader(layer.qweight, ref_math.pack_cols(q, 4, ref_math.AWQ_NIBBLE_OF_COLUMN), i)
        layer.qzeros.weight_loader(layer.qzeros, ref_math.pack_cols(z, 4, ref_math.AWQ_NIBBLE_OF_COLUMN), i)
        layer.scales.weight_loader(layer.scales, s, i)
        ws.append(ref_math.awq_reference_weight(q, z, s, 128, dt))
    finish(layer)
    x = torch.randn((m, k), generator=g).to(dt)
    assert (getattr(layer, "awq_marlin_kn", None) is not None) == (k % 256 == 0)
    out, _ = layer(x.to(gpu_device))  # unrepacked, m = 300 takes the dequantize + matmul branch
    assert ref_math.compute_max_diff(out.cpu(), x.float() @ torch.cat(ws, 1).float()) < 5e-3


def test_legacy_marlin_method(gpu_device):
    k, parts, dt, m = 512, [256, 128], torch.float16, 6
    layer = make_layer("marlin", dict(group_size=128), k, parts, dt, gpu_device)
    g = torch.Generator().manual_seed(3)
    w = torch.randn((k, sum(parts)), generator=g).to(dt)
    w_ref, mq, ms, _, _, _ = ref_math.marlin_quantize(w, 4, 128, False)
    layer.B.data.copy_(mq)  # Marlin-serialised checkpoints are stored pre-tiled
    layer.s.data.copy_(ms)
    finish(layer)
    x = torch.randn((m, k), generator=g).to(dt)
    out, _ = layer(x.to(gpu_device))
    assert ref_math.compute_max_diff(out.cpu(), x.float() @ w_ref.float()) < 6e-3


@pytest.mark.parametrize("serialized,scheme", [(False, "dynamic"), (True, "dynamic"), (True, "static")])
def test_fp8_method(gpu_device, serialized, scheme):
    k, parts, dt, m = 512, [256, 128], torch.bfloat16, 7
    cfg = dict(quant_method="fp8" if serialized else "none", activation_scheme=scheme)
    layer = make_layer("fp8", cfg, k, parts, dt, gpu_device)
    assert not layer.quant_method.use_marlin and layer.quant_method.cutlass_fp8_supported
    g = torch.Generator().manual_seed(4)
    ws = []
    x = torch.randn((m, k), generator=g).to(dt)
    for i, n in enumerate(parts):
        w = (torch.randn((n, k), generator=g) * 0.05).to(dt)  # [out, in]
        if serialized:
            sc = w.float().abs().max() / 448.0
            wq = (w.float() / sc).clamp(-448, 448).to(torch.float8_e4m3fn)
            layer.weight.weight_loader(layer.weight, wq, i)
            layer.weight_scale.weight_loader(layer.weight_scale, sc.reshape(()), i)
            if scheme == "static":
                layer.input_scale.weight_loader(layer.input_scale, (x.float().abs().max() / 448.0).reshape(()), i)
            ws.append(wq.float() * sc)
        else:
            layer.weight.weight_loader(layer.weight, w, i)
            ws.append(w.float())
    finish(layer)
    assert layer.weight.dtype == torch.float8_e4m3fn and layer.weight.stride(0) == 1  # column-major B
    out, _ = layer(x.to(gpu_device))
    ref = x.float() @ torch.cat(ws, 0).t()
    assert ref_math.compute_max_diff(out.cpu(), ref) < 6e-2  # fp8 activations + requantised weights


@pytest.mark.parametrize("strategy,dynamic", [("channel", True), ("tensor", True), ("tensor", False)])
def test_compressed_tensors_w8a8(gpu_device, strategy, dynamic):
    k, parts, dt, m = 512, [256, 128], torch.bfloat16, 11
    cfg = {"format": "int-quantized",
           "config_groups": {"g": {"targets": ["Linear"],
                                   "weights": {"num_bits": 8, "type": "int", "symmetric": True, "strategy": strategy},
                                   "input_activations": {"num_bits": 8, "type": "int", "symmetric": True,
                                                         "dynamic": dynamic, "strategy": "token" if dynamic else "tensor"}}}}
    layer = make_layer("compressed-tensors", cfg, k, parts, dt, gpu_device)
    g = torch.Generator().manual_seed(5)
    x = torch.randn((m, k), generator=g).to(dt)
    ws = []
    for i, n in enumerate(parts):
        w = torch.randn((n, k), generator=g) * 0.05
        if strategy == "channel":
            sc = w.abs().amax(dim=1, keepdim=True) / 127.0
        else:
            sc = (w.abs().max() / 127.0).reshape(1)
        wq = torch.round(w / sc).clamp(-128, 127).to(torch.int8)
        layer.weight.weight_loader(layer.weight, wq, i)
        layer.weight_scale.weight_loader(layer.weight_scale, sc.float(), i)
        ws.append(wq.float() * sc)
    if not dynamic:
        layer.input_scale.data.copy_((x.float().abs().max() / 127.0).reshape(1))
    finish(layer)
    out, _ = layer(x.to(gpu_device))
    ref = x.float() @ torch.cat(ws, 0).t()
    assert ref_math.compute_max_diff(out.cpu(), ref) < 3e-2  # int8 activation quantisation error


def test_compressed_tensors_wna16(gpu_device):
    k, parts, dt, m = 512, [256, 128], torch.bfloat16, 4
    cfg = {"format": "pack-quantized",
           "config_groups": {"g": {"targets": ["Linear"],
                                   "weights": {"num_bits": 4, "type": "int", "symmetric": True,
                                               "strategy": "group", "group_size": 128},
                                   "input_activations": None}}}
    layer = make_layer("compressed-tensors", cfg, k, parts, dt, gpu_device)
    g = torch.Generator().manual_seed(6)
    ws = []
    for i, n in enumerate(parts):
        w = torch.randn((k, n), generator=g).to(dt)
        w_ref, q, s, _, _ = ref_math.quantize_weights(w, 4, 128, False)
        # weight_packed [N, K/8]: consecutive K per int32 (compressed_tensors_wNa16.py:57-75)
        layer.weight_packed.weight_loader(layer.weight_packed, ref_math.gptq_pack(q, 4, k, n).t().contiguous(), i)
        layer.weight_scale.weight_loader(layer.weight_scale, s.t().contiguous(), i)
        ws.append(w_ref.to(dt))
    finish(layer)
    x = torch.randn((m, k), generator=g).to(dt)
    out, _ = layer(x.to(gpu_device))
    assert ref_math.compute_max_diff(out.cpu(), x.float() @ torch.cat(ws, 1).float()) < 6e-3
