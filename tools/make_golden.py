"""Generate tests/golden/*.npz from the REFERENCE itself.  Runs only in the build container.

Two sources:
  * the reference's CPU kernels (csrc/cpu/*.cpp) compiled into oracle/_ref/_C_ref.so by
    oracle/build_ref.py  -> paged_attention_v1/v2, reshape_and_cache, copy_blocks, rms_norm,
    fused_add_rms_norm, rotary_embedding, silu_and_mul / gelu_*;
  * the reference's Python quantisation utilities imported from /root/reference
    (quant_utils.py, marlin_utils.py, marlin_perms.py, gptq_marlin.py) -> quantize_weights,
    gptq_pack, marlin_weights, marlin_permute_scales, marlin perm tables.

Inputs come from tests/helpers.py (CPU RNG, explicit seeds).  Big inputs (KV caches) are NOT
stored: the fixture keeps the recipe arguments plus a sha256 of the regenerated input bytes, and
the reference OUTPUTS in full.  Fixtures are data only -- no reference source text is stored.

usage:  python tools/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLD = os.path.join(ROOT, "tests", "golden")

import helpers  # noqa: E402
from oracle import build_ref  # noqa: E402


def load_reference_python():
    sys.modules.setdefault("cpuinfo", types.ModuleType("cpuinfo"))  # optional dep, absent here
    sys.path.insert(0, build_ref.REF_ROOT)
    from vllm.model_executor.layers.quantization import gptq_marlin  # noqa: F401
    from vllm.model_executor.layers.quantization.utils import (marlin_perms, marlin_utils,
                                                               quant_utils)
    return quant_utils, marlin_utils, marlin_perms, gptq_marlin


def save(name, **arrays):
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


PA_CASES = [
    # name, seed, seqs, (q heads, kv heads), head, block, dtype, seq_lens, alibi
    ("pa_bf16_gqa4_ragged", 0, 8, (8, 2), 128, 16, torch.bfloat16,
     [1, 15, 16, 17, 511, 512, 513, 1500], False),
    ("pa_bf16_mha_alibi", 1, 4, (4, 4), 64, 16, torch.bfloat16, [3, 64, 700, 1025], True),
    ("pa_f32path_gqa8", 2, 3, (8, 1), 96, 16, torch.bfloat16, [129, 1, 640], False),
]


def gen_paged_attention():
    ops = torch.ops._C_ref
    for name, seed, ns, heads, hs, bs, dt, lens, alibi in PA_CASES:
        inp = helpers.make_paged_attention_inputs(seed, ns, heads, hs, bs, dt, seq_lens=lens,
                                                  num_blocks=256, use_alibi=alibi)
        q, kc, vc = inp["query"], inp["key_cache"], inp["value_cache"]
        out1 = torch.empty_like(q)
        ops.paged_attention_v1(out1, q, kc, vc, heads[1], inp["scale"], inp["block_tables"],
                               inp["seq_lens"], bs, inp["max_seq_len"], inp["alibi_slopes"],
                               "auto", 1.0, 0, 0, 0, 64, 0)
        mp = (inp["max_seq_len"] + 511) // 512
        out2 = torch.empty_like(q)
        tmp = torch.zeros((ns, heads[0], mp, hs), dtype=dt)
        es = torch.zeros((ns, heads[0], mp), dtype=torch.float32)
        ml = torch.zeros_like(es)
        ops.paged_attention_v2(out2, es, ml, tmp, q, kc, vc, heads[1], inp["scale"],
                               inp["block_tables"], inp["seq_lens"], bs, inp["max_seq_len"],
                               inp["alibi_slopes"], "auto", 1.0, 0, 0, 0, 64, 0)
        save(name, seed=seed, num_seqs=ns, num_q_heads=heads[0], num_kv_heads=heads[1],
             head_size=hs, block_size=bs, seq_lens=np.array(lens), use_alibi=alibi,
             input_sha=np.array(helpers.tensor_sha(q, kc, vc, inp["block_tables"])),
             out_v1=helpers.to_np(out1), out_v2=helpers.to_np(out2))


def gen_cache():
    cops = torch.ops._C_ref_cache_ops
    for name, seed, nt, nh, hs, bs, nb in [("rc_bf16_h8_d128_b16", 0, 42, 8, 128, 16, 64),
                                           ("rc_bf16_h2_d80_b16", 1, 17, 2, 80, 16, 32)]:
        inp = helpers.make_reshape_and_cache_inputs(seed, nt, nh, hs, bs, nb, torch.bfloat16)
        kc, vc = inp["key_cache"].clone(), inp["value_cache"].clone()
        cops.reshape_and_cache(inp["key"], inp["value"], kc, vc, inp["slot_mapping"], "auto", 1.0)
        # only the touched slots are stored: (block, offset) rows of both caches
        slots = inp["slot_mapping"]
        blk, off = slots // bs, slots % bs
        save(name, seed=seed, num_tokens=nt, num_heads=nh, head_size=hs, block_size=bs,
             num_blocks=nb, input_sha=np.array(helpers.tensor_sha(inp["key"], inp["value"],
                                                                  inp["key_cache"])),
             cache_sha=np.array(helpers.tensor_sha(kc, vc)),
             k_rows=helpers.to_np(kc[blk, :, :, off, :]), v_rows=helpers.to_np(vc[blk, :, :, off]))
    # copy_blocks
    inp = helpers.make_reshape_and_cache_inputs(3, 4, 4, 64, 16, 32, torch.bfloat16)
    kc, vc = inp["key_cache"].clone(), inp["value_cache"].clone()
    mapping = torch.tensor([[0, 5], [0, 9], [3, 7], [30, 1]], dtype=torch.int64)
    cops.copy_blocks([kc], [vc], mapping)
    save("copy_blocks_bf16", seed=3, mapping=mapping.numpy(),
         cache_sha=np.array(helpers.tensor_sha(kc, vc)))


def gen_glue():
    ops = torch.ops._C_ref
    g = torch.Generator().manual_seed(0)
    dt = torch.bfloat16
    x = torch.randn((7, 1024), generator=g).to(dt)
    res = torch.randn((7, 1024), generator=g).to(dt)
    w = (1 + 0.1 * torch.randn(1024, generator=g)).to(dt)
    out = torch.empty_like(x)
    ops.rms_norm(out, x, w, 1e-5)
    x2, r2 = x.clone(), res.clone()
    ops.fused_add_rms_norm(x2, r2, w, 1e-5)
    gu = torch.randn((5, 2 * 768), generator=g).to(dt)
    silu = torch.empty((5, 768), dtype=dt)
    ops.silu_and_mul(silu, gu)
    gelu = torch.empty((5, 768), dtype=dt)
    ops.gelu_and_mul(gelu, gu)
    gelut = torch.empty((5, 768), dtype=dt)
    ops.gelu_tanh_and_mul(gelut, gu)
    # rotary: neox + gptj, head 128, rot 128, 4 q heads / 2 kv heads
    hs, rot, maxpos = 128, 128, 512
    inv = 1.0 / (10000**(torch.arange(0, rot, 2).float() / rot))
    fr = torch.einsum("i,j->ij", torch.arange(maxpos).float(), inv)
    cache = torch.cat((fr.cos(), fr.sin()), dim=-1).to(dt)
    pos = torch.randint(0, maxpos, (11, ), generator=g)
    q = torch.randn((11, 4 * hs), generator=g).to(dt)
    k = torch.randn((11, 2 * hs), generator=g).to(dt)
    qn, kn = q.clone(), k.clone()
    ops.rotary_embedding(pos, qn, kn, hs, cache, True)
    qj, kj = q.clone(), k.clone()
    ops.rotary_embedding(pos, qj, kj, hs, cache, False)
    save("glue_bf16", x=helpers.to_np(x), res=helpers.to_np(res), w=helpers.to_np(w),
         rms=helpers.to_np(out), fused_x=helpers.to_np(x2), fused_res=helpers.to_np(r2),
         gate_up=helpers.to_np(gu), silu=helpers.to_np(silu), gelu=helpers.to_np(gelu),
         gelu_tanh=helpers.to_np(gelut), rope_cache=helpers.to_np(cache), rope_pos=pos.numpy(),
         rope_q=helpers.to_np(q), rope_k=helpers.to_np(k), rope_q_neox=helpers.to_np(qn),
         rope_k_neox=helpers.to_np(kn), rope_q_gptj=helpers.to_np(qj),
         rope_k_gptj=helpers.to_np(kj))


def gen_quant():
    quant_utils, marlin_utils, marlin_perms, gptq_marlin = load_reference_python()
    # permutation tables
    save("marlin_perms", perm4=marlin_perms.marlin_perm[4].numpy(),
         perm8=marlin_perms.marlin_perm[8].numpy(),
         scale_perm=np.array(marlin_perms.marlin_scale_perm[4]),
         scale_perm_single=np.array(marlin_perms.marlin_scale_perm_single[4]))
    for name, seed, k, n, bits, gs in [("mq_k256_n128_b4_g128", 0, 256, 128, 4, 128),
                                       ("mq_k128_n64_b4_gm1", 1, 128, 64, 4, -1),
                                       ("mq_k256_n192_b4_g32", 2, 256, 192, 4, 32),
                                       ("mq_k128_n128_b8_g64", 3, 128, 128, 8, 64)]:
        g = torch.Generator().manual_seed(seed)
        w = torch.randn((k, n), generator=g).half()
        gsz = k if gs == -1 else gs
        w_ref, q_w, s, g_idx, rand_perm = quant_utils.quantize_weights(w, bits, gsz, False)
        packed = quant_utils.gptq_pack(q_w, bits, k, n)
        mw = marlin_utils.marlin_weights(q_w, k, n, bits, marlin_perms.marlin_perm[bits])
        ms = marlin_utils.marlin_permute_scales(s, k, n, gsz, marlin_perms.marlin_scale_perm[bits],
                                                marlin_perms.marlin_scale_perm_single[bits])
        ms2 = gptq_marlin.marlin_permute_scales(s, k, n, gsz, bits)
        assert torch.equal(ms, ms2)
        # an act-order style repack: rows permuted by a fixed permutation
        perm = torch.randperm(k, generator=g).to(torch.int32)
        mw_perm = marlin_utils.marlin_weights(q_w[perm.long()], k, n, bits,
                                              marlin_perms.marlin_perm[bits])
        save(name, seed=seed, size_k=k, size_n=n, num_bits=bits, group_size=gs,
             w=helpers.to_np(w), w_ref=helpers.to_np(w_ref), q_w=q_w.numpy().astype(np.uint8),
             s=helpers.to_np(s), gptq_packed=packed.numpy(), marlin_q_w=mw.numpy(),
             marlin_s=helpers.to_np(ms), perm=perm.numpy(), marlin_q_w_perm=mw_perm.numpy())


def gen_param_tables():
    """parameter names / shapes / dtypes / sharding attributes the reference's LinearMethods create
    for the Llama-3-8B qkv projection (last row)."""
    import json
    load_reference_python()
    from vllm.model_executor.layers.quantization import QUANTIZATION_METHODS
    cfgs = {
        "gptq_marlin": dict(bits=4, group_size=128, desc_act=False, sym=True),
        "gptq_marlin_act_order": dict(bits=4, group_size=128, desc_act=True, sym=True),
        "gptq": dict(bits=4, group_size=128, desc_act=False),
        "awq": dict(w_bit=4, q_group_size=128, zero_point=True),
        "marlin": dict(group_size=128),
        "fp8": dict(quant_method="fp8", activation_scheme="static"),
    }
    keep = ("input_dim", "output_dim", "packed_dim", "pack_factor", "marlin_tile_size",
            "needs_scalar_to_array")
    out = {}

    class Dummy(torch.nn.Module):
        pass

    for name, cfg in cfgs.items():
        method_name = name.replace("_act_order", "")
        qc = QUANTIZATION_METHODS[method_name].from_config(cfg)
        lm_cls = type(qc.get_quant_method.__func__) if False else None
        # build the linear method directly (get_quant_method needs a LinearBase instance)
        import importlib
        mod = importlib.import_module(type(qc).__module__)
        layer = Dummy()
        try:
            lm = [getattr(mod, n) for n in dir(mod)
                  if n.endswith("LinearMethod") and n != "LinearMethodBase"][0](qc)
            lm.create_weights(layer, 4096, [4096, 1024, 1024], 4096, 6144,
                              torch.float16 if name in ("gptq", "marlin") else torch.bfloat16,
                              weight_loader=None)
        except RuntimeError as e:  # e.g. the legacy marlin method allocates its workspace on "cuda"
            print(f"  (skipped {name}: {str(e)[:60]})")
            continue
        table = {}
        for pname, prm in layer.named_parameters():
            attrs = {k: (str(getattr(prm, k)) if k == "pack_factor" else getattr(prm, k))
                     for k in keep if hasattr(prm, k)}
            table[pname] = dict(shape=list(prm.shape), dtype=str(prm.dtype), device=prm.device.type,
                                attrs=attrs)
        out[name] = table
    with open(os.path.join(GOLD, "linear_method_params.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("  linear_method_params.json")


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    build_ref.build()
    assert build_ref.load_ref(), "oracle/_ref not loadable"
    gen_paged_attention()
    gen_cache()
    gen_glue()
    gen_quant()
    gen_param_tables()
    print("done")


