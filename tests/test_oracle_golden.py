"""CPU: the oracle (oracle/oracle.c, oracle/ref_math.py) against the golden vectors that
tools/make_golden.py produced by running the reference itself.  This is what pins parity."""
import glob
import os

import numpy as np
import pytest
import torch

import helpers
import oracle
from oracle import ref_math

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BF16 = torch.bfloat16


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


PA_NAMES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "pa_*.npz")))


def pa_inputs(g):
    return helpers.make_paged_attention_inputs(
        int(g["seed"]), int(g["num_seqs"]), (int(g["num_q_heads"]), int(g["num_kv_heads"])),
        int(g["head_size"]), int(g["block_size"]), BF16, seq_lens=[int(v) for v in g["seq_lens"]],
        num_blocks=256, use_alibi=bool(g["use_alibi"]))


@pytest.mark.parametrize("name", PA_NAMES)
def test_paged_attention_oracle_vs_reference(name):
    g = gold(name)
    inp = pa_inputs(g)
    assert helpers.tensor_sha(inp["query"], inp["key_cache"], inp["value_cache"],
                              inp["block_tables"]) == str(g["input_sha"]), "input recipe drifted"
    ref1 = helpers.from_np(g["out_v1"], BF16).float()
    ref2 = helpers.from_np(g["out_v2"], BF16).float()
    args = (inp["query"], inp["key_cache"], inp["value_cache"], inp["num_kv_heads"], inp["scale"],
            inp["block_tables"], inp["seq_lens"], inp["block_size"])
    o1 = oracle.paged_attention(*args, alibi_slopes=inp["alibi_slopes"]).float()
    o2, es, ml, tmp = oracle.paged_attention(*args, alibi_slopes=inp["alibi_slopes"],
                                             partition_size=512)
    # bf16 outputs of magnitude <= ~0.1: one bf16 ulp there is < 5e-4 (reference CPU kernel
    # accumulates in bf16-vector order, the oracle in fp32)
    assert torch.allclose(o1, ref1, atol=1e-3, rtol=1e-2)
    assert torch.allclose(o2.float(), ref2, atol=1e-3, rtol=1e-2)
    # and against the plain-torch checker of the reference's own test
    t = helpers.ref_paged_attention_torch(inp)
    assert torch.allclose(o1, t, atol=1e-3, rtol=1e-2)


@pytest.mark.parametrize("name", ["rc_bf16_h8_d128_b16", "rc_bf16_h2_d80_b16"])
def test_reshape_and_cache_bit_exact(name):
    g = gold(name)
    bs = int(g["block_size"])
    inp = helpers.make_reshape_and_cache_inputs(int(g["seed"]), int(g["num_tokens"]),
                                                int(g["num_heads"]), int(g["head_size"]), bs,
                                                int(g["num_blocks"]), BF16)
    assert helpers.tensor_sha(inp["key"], inp["value"], inp["key_cache"]) == str(g["input_sha"])
    kc, vc = inp["key_cache"].clone(), inp["value_cache"].clone()
    oracle.reshape_and_cache(inp["key"], inp["value"], kc, vc, inp["slot_mapping"])
    assert helpers.tensor_sha(kc, vc) == str(g["cache_sha"])  # whole caches, bit for bit
    blk, off = inp["slot_mapping"] // bs, inp["slot_mapping"] % bs
    assert np.array_equal(helpers.to_np(kc[blk, :, :, off, :]), g["k_rows"])
    assert np.array_equal(helpers.to_np(vc[blk, :, :, off]), g["v_rows"])


def test_copy_blocks_bit_exact():
    g = gold("copy_blocks_bf16")
    inp = helpers.make_reshape_and_cache_inputs(int(g["seed"]), 4, 4, 64, 16, 32, BF16)
    kc, vc = inp["key_cache"].clone(), inp["value_cache"].clone()
    oracle.copy_blocks(kc, vc, torch.from_numpy(g["mapping"]))
    assert helpers.tensor_sha(kc, vc) == str(g["cache_sha"])


def test_glue_ops():
    g = gold("glue_bf16")
    T = lambda k: helpers.from_np(g[k], BF16)  # noqa: E731
    x, res, w = T("x"), T("res"), T("w")
    out = oracle.rms_norm(x, w, 1e-5)
    # the reference CPU kernel keeps x*rsqrt in fp32 before the weight multiply while the CUDA
    # kernel (and the oracle) round to bf16 first (layernorm_kernels.cu:41-42): allow 1 bf16 ulp
    assert torch.allclose(out.float(), T("rms").float(), atol=2e-2, rtol=1.6e-2)
    r2 = res.clone()
    out2 = oracle.rms_norm(x.clone(), w, 1e-5, residual=r2)
    # the reference's CPU build here (gcc 11, no avx512_bf16) converts fp32->bf16 by
    # TRUNCATION (csrc/cpu/cpu_types_x86.hpp BF16Vec16(FP32Vec16)); the CUDA kernel and the
    # oracle round to nearest even -> results agree to 1 bf16 ulp, not bit for bit
    assert torch.allclose(r2.float(), T("fused_res").float(), atol=0, rtol=2**-7)
    assert torch.allclose(out2.float(), T("fused_x").float(), atol=2e-2, rtol=1.6e-2)
    for act, key in ((0, "silu"), (1, "gelu"), (2, "gelu_tanh")):
        o = oracle.act_and_mul(T("gate_up"), act)
        assert torch.allclose(o.float(), T(key).float(), atol=1e-2, rtol=1.6e-2), key
    pos = torch.from_numpy(g["rope_pos"])
    for neox, tag in ((True, "neox"), (False, "gptj")):
        q, k = T("rope_q").clone(), T("rope_k").clone()
        oracle.rotary_embedding(pos, q, k, 128, T("rope_cache"), neox)
        assert torch.allclose(q.float(), T("rope_q_" + tag).float(), atol=2e-2, rtol=1.6e-2)
        assert torch.allclose(k.float(), T("rope_k_" + tag).float(), atol=2e-2, rtol=1.6e-2)


def test_marlin_perm_tables():
    g = gold("marlin_perms")
    assert np.array_equal(ref_math.marlin_perm(4).numpy(), g["perm4"])
    assert np.array_equal(ref_math.marlin_perm(8).numpy(), g["perm8"])
    grouped, single = ref_math.scale_perms()
    assert grouped == list(g["scale_perm"]) and single == list(g["scale_perm_single"])


MQ_NAMES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "mq_*.npz")))


@pytest.mark.parametrize("name", MQ_NAMES)
def test_quantize_pack_repack_bit_exact(name):
    g = gold(name)
    k, n, bits, gs = int(g["size_k"]), int(g["size_n"]), int(g["num_bits"]), int(g["group_size"])
    w = helpers.from_np(g["w"], torch.float16)
    w_ref, q_w, s, _, _ = ref_math.quantize_weights(w, bits, gs, False)
    assert np.array_equal(q_w.numpy().astype(np.uint8), g["q_w"])
    assert np.array_equal(helpers.to_np(s), g["s"])
    assert np.array_equal(helpers.to_np(w_ref), g["w_ref"])
    packed = ref_math.gptq_pack(q_w, bits, k, n)
    assert np.array_equal(packed.numpy(), g["gptq_packed"])
    assert torch.equal(ref_math.gptq_unpack(packed, bits), q_w)
    # python restatement of the marlin layout
    assert np.array_equal(ref_math.marlin_weights(q_w, k, n, bits).numpy(), g["marlin_q_w"])
    gsz = k if gs == -1 else gs
    assert np.array_equal(helpers.to_np(ref_math.marlin_permute_scales(s, k, n, gsz)), g["marlin_s"])
    # C restatement of gptq_marlin_repack (GPTQ -> Marlin), without and with a row permutation
    mw = oracle.gptq_marlin_repack(packed, None, k, n, bits)
    assert np.array_equal(mw.numpy(), g["marlin_q_w"])
    perm = torch.from_numpy(g["perm"])
    mwp = oracle.gptq_marlin_repack(packed, perm, k, n, bits)
    assert np.array_equal(mwp.numpy(), g["marlin_q_w_perm"])
    # and the inverse used by the GEMM oracle
    assert np.array_equal(oracle.marlin_unpack(mw, k, n, bits).numpy(), g["q_w"])


@pytest.mark.parametrize("name", [n for n in MQ_NAMES])
def test_gemm_oracle_matches_a_at_w_ref(name):
    """oracle GEMM == a @ w_ref (tests/kernels/test_marlin_gemm.py:172) on the golden weights."""
    g = gold(name)
    k, n, bits = int(g["size_k"]), int(g["size_n"]), int(g["num_bits"])
    gen = torch.Generator().manual_seed(5)
    a = torch.randn((13, k), generator=gen).half()
    w_ref = helpers.from_np(g["w_ref"], torch.float16)
    ms = helpers.from_np(g["marlin_s"], torch.float16)
    c = oracle.gptq_marlin_gemm(a, torch.from_numpy(g["marlin_q_w"]), ms, None, None, bits, 13, n, k)
    ref = (a.float() @ w_ref.float())
    assert ref_math.compute_max_diff(c, ref) < 1e-3


# ---------------------------------------------------------------------------------------------
# tests/ref_llama.py (the fp32 torch model the GPU end-to-end tests compare with) pinned to the
# REFERENCE's own LlamaForCausalLM run on CPU (tests/golden/tiny_llama_*.npz, tools/make_golden_model.py)
@pytest.mark.parametrize("case", ["bf16", "w4a16"])
def test_ref_llama_pinned_to_reference_model(case):
    import types
    from ref_llama import RefLlama
    g = gold(f"tiny_llama_{case}")
    ckpt = helpers.tiny_llama_checkpoint(int(g["seed"]), BF16)
    assert helpers.tensor_sha(*[ckpt[k] for k in sorted(ckpt)]) == str(g["ckpt_sha"]), "checkpoint recipe drifted"
    weights = helpers.gptq_checkpoint_from_dense(ckpt, 4, 128) if case == "w4a16" else ckpt
    a = helpers.TINY_LLAMA
    arch = types.SimpleNamespace(**a, head_dim=a["hidden_size"] // a["num_attention_heads"])
    ref = RefLlama(arch, weights)
    prompts = torch.from_numpy(g["prompts"])
    tokens = torch.from_numpy(g["tokens"])
    steps = int(g["steps"])
    # the reference ran bf16 end to end, ref_llama keeps fp32 activations: bf16-level agreement (3e-2 of
    # the mean |logit|, the tolerance the GPU end-to-end tests state)
    for b in range(prompts.shape[0]):
        seq = prompts[b].tolist()
        for s in range(steps + 1):
            logits = ref.forward(torch.tensor(seq))[-1]
            want = torch.from_numpy(g["prompt_logits"][b] if s == 0 else g["step_logits"][s - 1][b])
            rel = ((logits - want).abs().mean() / want.abs().mean()).item()
            assert rel < 3e-2, (case, b, s, rel)
            if float(g["top2_margin"][s][b]) > 4 * float((logits - want).abs().max()):
                assert int(logits.argmax()) == int(tokens[s][b]), (case, b, s)
            seq.append(int(tokens[s][b]))   # teacher forcing with the reference's tokens


BSA_NAMES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "bsa_*.npz")))


def bsa_case(name):
    """inputs (seeded recipe), sparse parameters and the output of the reference's own checker
    (tools/make_golden_blocksparse.py: ref_single_query_cached_kv_attention of
    tests/kernels/test_blocksparse_attention.py executed in place)"""
    g = gold(name)
    seed, nseq, nq, nkv, hs, bs, alibi = (int(v) for v in g["recipe"])
    inp = helpers.make_paged_attention_inputs(seed, nseq, (nq, nkv), hs, bs, BF16,
                                              seq_lens=[int(v) for v in g["seq_lens"]], num_blocks=256,
                                              use_alibi=bool(alibi))
    keys = ("tp_rank", "local_blocks", "vert_stride", "block_size", "head_sliding_step")
    sparse = dict(zip(keys, (int(v) for v in g["sparse"])))
    return inp, sparse, torch.from_numpy(g["out"]).float()


@pytest.mark.parametrize("name", BSA_NAMES)
def test_blocksparse_checker_vs_reference(name):
    """the Python checker of block-sparse paged attention (tests/helpers.py) against the reference test's own"""
    assert len(BSA_NAMES) == 3
    inp, sparse, ref = bsa_case(name)
    out = helpers.ref_paged_attention_torch(inp, blocksparse=sparse)
    # the reference checker rounds the probabilities and the output to bf16 (measured: max abs 3e-4, mean 0.2 %)
    assert torch.allclose(out, ref, atol=5e-4, rtol=1e-2), (out - ref).abs().max()
    assert ((out - ref).abs().mean() / ref.abs().mean()).item() < 1e-2
    # and the mask does something: the dense result is 30-75 % away
    dense = helpers.ref_paged_attention_torch(inp)
    assert ((dense - ref).abs().mean() / ref.abs().mean()).item() > 0.1
