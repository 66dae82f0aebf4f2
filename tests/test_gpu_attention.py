"""GPU parity: paged_attention_v1/v2 (HIP, through torch.ops._C -> C ABI) vs the CPU oracle and
the golden vectors of the reference.  Recipe and tolerances follow the reference's
tests/kernels/test_attention.py:119-284 (atol 1e-3 / rtol 1e-5; fp8 KV: atol 1e-2)."""
import os

import numpy as np
import pytest
import torch

import helpers
import oracle

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def run_hip(inp, version, dev, kv_cache_dtype="auto", kv_scale=1.0, query=None, sparse=None):
    """sparse = dict(tp_rank, local_blocks, vert_stride, block_size, head_sliding_step): block-sparse attention"""
    from neural_magic_vllm_amd import _custom_ops as ops
    sp = (0, 0, 0, 64, 0) if sparse is None else tuple(sparse[k] for k in (
        "tp_rank", "local_blocks", "vert_stride", "block_size", "head_sliding_step"))
    q = inp["query"].to(dev) if query is None else query
    kc, vc = inp["key_cache"].to(dev), inp["value_cache"].to(dev)
    bt, sl = inp["block_tables"].to(dev), inp["seq_lens"].to(dev)
    al = None if inp["alibi_slopes"] is None else inp["alibi_slopes"].to(dev)
    out = torch.full(tuple(q.shape), float("nan"), dtype=q.dtype, device=dev)
    nkv, bs, msl = inp["num_kv_heads"], inp["block_size"], inp["max_seq_len"]
    if version == "v1":
        ops.paged_attention_v1(out, q, kc, vc, nkv, inp["scale"], bt, sl, bs, msl, al,
                               kv_cache_dtype, kv_scale, *sp)
        return out.cpu(), None
    ns, nh, hs = q.shape
    mp = (msl + 511) // 512
    tmp = torch.empty((ns, nh, mp, hs), dtype=q.dtype, device=dev)
    es = torch.empty((ns, nh, mp), dtype=torch.float32, device=dev)
    ml = torch.empty_like(es)
    ops.paged_attention_v2(out, es, ml, tmp, q, kc, vc, nkv, inp["scale"], bt, sl, bs, msl, al,
                           kv_cache_dtype, kv_scale, *sp)
    return out.cpu(), (es.cpu(), ml.cpu(), tmp.cpu())


def run_oracle(inp, kv_cache_dtype="auto", kv_scale=1.0, partition_size=0):
    return oracle.paged_attention(inp["query"], inp["key_cache"], inp["value_cache"],
                                  inp["num_kv_heads"], inp["scale"], inp["block_tables"],
                                  inp["seq_lens"], inp["block_size"],
                                  alibi_slopes=inp["alibi_slopes"], kv_cache_dtype=kv_cache_dtype,
                                  kv_scale=kv_scale, partition_size=partition_size)


_LAST = {}


def inputs_and_ref(key, make, **oracle_kw):
    """the v1 and v2 cases of a parameter set run back to back on identical seeded inputs: build the
    inputs and the (CPU, slow) oracle result once for the pair"""
    if _LAST.get("key") != key:
        inp = make()
        _LAST.update(key=key, inp=inp, ref=run_oracle(inp, **oracle_kw))
    return _LAST["inp"], _LAST["ref"]


def check(out, ref, atol=1e-3, rtol=1e-5):
    out, ref = out.float(), ref.float()
    assert not torch.isnan(out).any(), "NaN / unwritten output"
    err = (out - ref).abs().max().item()
    assert torch.allclose(out, ref, atol=atol, rtol=rtol), f"max abs err {err}"
    # systematic-error guard, much tighter than the elementwise tolerance
    rel = ((out - ref).abs().mean() / ref.abs().mean().clamp_min(1e-9)).item()
    assert rel < 2e-2, f"mean relative error {rel}"


@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("num_heads", [(40, 40), (64, 8), (32, 8)])
@pytest.mark.parametrize("head_size", [64, 80, 96, 112, 128, 192, 256])
@pytest.mark.parametrize("block_size", [16, 32])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_paged_attention(gpu_device, version, num_heads, head_size, block_size, dtype):
    inp, ref = inputs_and_ref(("plain", num_heads, head_size, block_size, dtype),
                              lambda: helpers.make_paged_attention_inputs(0, 7, num_heads, head_size, block_size,
                                                                          dtype, max_seq_len=1800, num_blocks=512))
    out, _ = run_hip(inp, version, gpu_device)
    check(out, ref)


@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("num_heads", [(40, 40), (64, 8), (12, 6)])
@pytest.mark.parametrize("head_size", [64, 128])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_paged_attention_alibi(gpu_device, version, num_heads, head_size, dtype):
    inp, ref = inputs_and_ref(("alibi", num_heads, head_size, dtype),
                              lambda: helpers.make_paged_attention_inputs(1, 7, num_heads, head_size, 16, dtype,
                                                                          max_seq_len=1300, num_blocks=512,
                                                                          use_alibi=True))
    out, _ = run_hip(inp, version, gpu_device)
    check(out, ref)


@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("num_heads", [(40, 40), (64, 8), (32, 8)])
@pytest.mark.parametrize("head_size", [64, 80, 128, 256])
@pytest.mark.parametrize("block_size", [8, 16, 32])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
def test_paged_attention_fp8_kv(gpu_device, version, num_heads, head_size, block_size, dtype):
    inp, ref = inputs_and_ref(("fp8", num_heads, head_size, block_size, dtype),
                              lambda: helpers.make_paged_attention_inputs(2, 5, num_heads, head_size, block_size,
                                                                          dtype, max_seq_len=1100, num_blocks=384,
                                                                          kv_cache_dtype="fp8"),
                              kv_cache_dtype="fp8", kv_scale=1.5)
    out, _ = run_hip(inp, version, gpu_device, kv_cache_dtype="fp8", kv_scale=1.5)
    # same fp8 bytes on both sides: only accumulation order differs
    check(out, ref, atol=2e-3)


@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("head_size", [64, 128])
def test_paged_attention_block8(gpu_device, version, head_size):
    inp = helpers.make_paged_attention_inputs(3, 6, (32, 8), head_size, 8, torch.bfloat16,
                                              max_seq_len=900, num_blocks=768)
    out, _ = run_hip(inp, version, gpu_device)
    check(out, run_oracle(inp))


@pytest.mark.parametrize("seq_lens", [[1], [1, 2, 3], [15, 16, 17], [63, 64, 65], [511, 512, 513],
                                      [1024, 1025, 1], [256] * 9])
@pytest.mark.parametrize("version", ["v1", "v2"])
def test_paged_attention_edges(gpu_device, version, seq_lens):
    inp = helpers.make_paged_attention_inputs(4, len(seq_lens), (32, 8), 128, 16, torch.bfloat16,
                                              seq_lens=seq_lens, num_blocks=256)
    out, _ = run_hip(inp, version, gpu_device)
    check(out, run_oracle(inp))


@pytest.mark.parametrize("version", ["v1", "v2"])
def test_paged_attention_long_context(gpu_device, version):
    """v1 has no shared-memory limit here (the reference caps it at 8192 via smem)."""
    inp = helpers.make_paged_attention_inputs(5, 2, (32, 8), 128, 16, torch.bfloat16,
                                              seq_lens=[12000, 9001], num_blocks=1400)
    out, _ = run_hip(inp, version, gpu_device)
    check(out, run_oracle(inp))


def test_paged_attention_strided_query(gpu_device):
    """q is a slice of the fused qkv output (query.stride(0) != heads*head_size)."""
    inp = helpers.make_paged_attention_inputs(6, 5, (32, 8), 128, 16, torch.bfloat16,
                                              max_seq_len=700, num_blocks=256)
    ns, nh, hs = inp["query"].shape
    qkv = torch.zeros((ns, (nh + 16) * hs), dtype=torch.bfloat16, device=gpu_device)
    qkv[:, :nh * hs] = inp["query"].reshape(ns, -1).to(gpu_device)
    q = qkv[:, :nh * hs].view(ns, nh, hs)
    assert q.stride(0) != nh * hs
    out, _ = run_hip(inp, "v1", gpu_device, query=q)
    check(out, run_oracle(inp))


def test_v2_partition_outputs(gpu_device):
    """exp_sums / max_logits / tmp_out keep the reference's meaning (attention_kernels.cu:350-361)."""
    inp = helpers.make_paged_attention_inputs(7, 3, (8, 2), 128, 16, torch.bfloat16,
                                              seq_lens=[1500, 513, 40], num_blocks=256)
    out, (es, ml, tmp) = run_hip(inp, "v2", gpu_device)
    o_ref, es_ref, ml_ref, tmp_ref = run_oracle(inp, partition_size=512)
    check(out, o_ref)
    for s, L in enumerate([1500, 513, 40]):
        npart = (L + 511) // 512
        assert torch.allclose(ml[s, :, :npart], ml_ref[s, :, :npart], atol=1e-4, rtol=1e-3)
        assert torch.allclose(es[s, :, :npart], es_ref[s, :, :npart], atol=1e-3, rtol=2e-3)
        assert torch.allclose(tmp[s, :, :npart].float(), tmp_ref[s, :, :npart].float(), atol=1e-3,
                              rtol=1e-2)


@pytest.mark.parametrize("name", ["pa_bf16_gqa4_ragged", "pa_bf16_mha_alibi", "pa_f32path_gqa8"])
@pytest.mark.parametrize("version", ["v1", "v2"])
def test_paged_attention_golden(gpu_device, name, version):
    """directly against what the reference's own CPU kernels produced (tests/golden)."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    inp = helpers.make_paged_attention_inputs(
        int(g["seed"]), int(g["num_seqs"]), (int(g["num_q_heads"]), int(g["num_kv_heads"])),
        int(g["head_size"]), int(g["block_size"]), torch.bfloat16,
        seq_lens=[int(v) for v in g["seq_lens"]], num_blocks=256, use_alibi=bool(g["use_alibi"]))
    assert helpers.tensor_sha(inp["query"], inp["key_cache"], inp["value_cache"],
                              inp["block_tables"]) == str(g["input_sha"])
    out, _ = run_hip(inp, version, gpu_device)
    ref = helpers.from_np(g["out_" + version], torch.bfloat16)
    check(out, ref, atol=1e-3, rtol=1e-2)


@pytest.mark.parametrize("kv_cache_dtype", ["auto", "fp8"])
def test_paged_attention_bench_size(gpu_device, kv_cache_dtype):
    """BASELINE.json configs[2] / [3] at full size: 64 sequences around 512 tokens (ragged: every residue of the
    4-wave x 64-token window split), Llama-3-8B heads (32 q / 8 kv, D = 128), block 16, against the oracle"""
    lens = [512 + (i * 7) % 130 for i in range(64)]
    lens[0], lens[1], lens[2] = 512, 513, 641
    inp = helpers.make_paged_attention_inputs(9, 64, (32, 8), 128, 16, torch.bfloat16, seq_lens=lens,
                                              num_blocks=64 * 41 + 8, kv_cache_dtype=kv_cache_dtype)
    kv_scale = 0.5 if kv_cache_dtype == "fp8" else 1.0
    ref = run_oracle(inp, kv_cache_dtype=kv_cache_dtype, kv_scale=kv_scale)
    for version in ("v1", "v2"):
        out, _ = run_hip(inp, version, gpu_device, kv_cache_dtype=kv_cache_dtype, kv_scale=kv_scale)
        check(out, ref, atol=2e-3 if kv_cache_dtype == "fp8" else 1e-3, rtol=1e-2)


@pytest.mark.parametrize("name", ["bsa_qslide", "bsa_kvslide_alibi", "bsa_homo"])
@pytest.mark.parametrize("version", ["v1", "v2"])
def test_blocksparse_paged_attention_golden(gpu_device, name, version):
    """block-sparse paged attention (attention_kernels.cu:209-251) against the output of the reference test's own
    checker (tests/golden/bsa_*.npz, tools/make_golden_blocksparse.py)"""
    import test_oracle_golden
    inp, sparse, ref = test_oracle_golden.bsa_case(name)
    out, _ = run_hip(inp, version, gpu_device, sparse=sparse)
    check(out, ref, atol=1e-3, rtol=1e-2)


@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("kv_cache_dtype", ["auto", "fp8"])
@pytest.mark.parametrize("sliding", [0, 2, -1])
@pytest.mark.parametrize("num_heads,head_size,block_size", [((40, 40), 64, 16), ((64, 8), 112, 32), ((32, 8), 128, 16)])
def test_blocksparse_paged_attention(gpu_device, version, kv_cache_dtype, sliding, num_heads, head_size, block_size):
    """the reference's parameter grid (tests/kernels/test_blocksparse_attention.py:33-45: 16 local blocks, vertical
    stride 8, sparsity blocks of 64 tokens, head sliding 0 / 2 / -1) against the Python checker; contexts long
    enough that whole 64-token windows and whole v2 partitions are masked for some heads"""
    sparse = dict(tp_rank=0, local_blocks=16, vert_stride=8, block_size=64, head_sliding_step=sliding)
    inp = helpers.make_paged_attention_inputs(5, 3, num_heads, head_size, block_size, torch.bfloat16,
                                              seq_lens=[3000, 1100, 70], num_blocks=512,
                                              kv_cache_dtype=kv_cache_dtype)
    kv_scale = 0.5 if kv_cache_dtype == "fp8" else 1.0
    ref = helpers.ref_paged_attention_torch(inp, kv_scale=kv_scale, blocksparse=sparse)
    out, _ = run_hip(inp, version, gpu_device, kv_cache_dtype=kv_cache_dtype, kv_scale=kv_scale, sparse=sparse)
    check(out, ref, atol=2e-3 if kv_cache_dtype == "fp8" else 1e-3, rtol=1e-2)
    # a small local window: most of a long context is masked, for some heads whole partitions
    sparse2 = dict(tp_rank=1, local_blocks=1, vert_stride=16, block_size=64, head_sliding_step=sliding)
    ref2 = helpers.ref_paged_attention_torch(inp, kv_scale=kv_scale, blocksparse=sparse2)
    out2, _ = run_hip(inp, version, gpu_device, kv_cache_dtype=kv_cache_dtype, kv_scale=kv_scale, sparse=sparse2)
    check(out2, ref2, atol=2e-3 if kv_cache_dtype == "fp8" else 1e-3, rtol=1e-2)


def test_unsupported_configs_raise(gpu_device):
    from neural_magic_vllm_amd import _custom_ops as ops
    inp = helpers.make_paged_attention_inputs(0, 2, (4, 4), 128, 16, torch.bfloat16,
                                              seq_lens=[10, 20], num_blocks=16)
    q = inp["query"].to(gpu_device)
    kc, vc = inp["key_cache"].to(gpu_device), inp["value_cache"].to(gpu_device)
    bt, sl = inp["block_tables"].to(gpu_device), inp["seq_lens"].to(gpu_device)
    out = torch.empty_like(q)
    with pytest.raises(RuntimeError, match="kv cache"):
        ops.paged_attention_v1(out, q, kc, vc, 4, 0.1, bt, sl, 16, 20, None, "fp8_e5m2", 1.0)
    with pytest.raises(RuntimeError, match="block size"):
        ops.paged_attention_v1(out, q, kc, vc, 4, 0.1, bt, sl, 24, 20, None, "auto", 1.0)
    q72 = torch.zeros((2, 4, 72), dtype=torch.bfloat16, device=gpu_device)
    with pytest.raises(RuntimeError, match="head size"):
        ops.paged_attention_v1(torch.empty_like(q72), q72, kc, vc, 4, 0.1, bt, sl, 16, 20, None,
                               "auto", 1.0)


@pytest.mark.parametrize("heads,kv_heads,head_size", [(32, 8, 128), (8, 8, 64), (16, 2, 128), (4, 1, 128)])
@pytest.mark.parametrize("kv_cache_dtype", ["auto", "fp8"])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
@pytest.mark.parametrize("num_seqs,max_ctx", [(5, 530), (3, 1300), (64, 40)])
def test_paged_attention_rope_partial_matches_separate_launches(gpu_device, heads, kv_heads, head_size,
                                                                kv_cache_dtype, dtype, num_seqs, max_ctx):
    """rope + cache write + paged attention (v1 and v2) in one launch, from fp32 split-K slabs of the qkv
    projection, against rotary_embedding_and_cache_partial + PagedAttention.forward_decode: attention
    output and both caches, bit for bit"""
    from neural_magic_vllm_amd import _custom_ops as ops
    from neural_magic_vllm_amd.attention.ops.paged_attn import PagedAttention
    d = gpu_device
    g = torch.Generator().manual_seed(11)
    n = (heads + 2 * kv_heads) * head_size
    block_size, max_pos = 16, 4096
    # positions: new token index per sequence (context = pos tokens already cached), incl. pos = 0
    pos = torch.randint(1, max_ctx, (num_seqs, ), generator=g)
    pos[0] = 0
    pos[-1] = max_ctx - 1
    seq_lens = (pos + 1).to(torch.int32)
    blocks_per_seq = (max_ctx + block_size - 1) // block_size
    num_blocks = num_seqs * blocks_per_seq + 3
    bt = torch.randperm(num_blocks, generator=g)[:num_seqs * blocks_per_seq].view(num_seqs, blocks_per_seq).to(torch.int32)
    slots = torch.tensor([int(bt[i, int(pos[i]) // block_size]) * block_size + int(pos[i]) % block_size
                          for i in range(num_seqs)])
    slab = (torch.randn((3, num_seqs, n), generator=g) * 0.5).to(d)          # three "splits"
    cos_sin = torch.randn((max_pos, head_size), generator=g).to(dtype).to(d)
    cdt = torch.uint8 if kv_cache_dtype == "fp8" else dtype
    x = 16 // torch.tensor([], dtype=cdt).element_size()
    kv_scale = 0.5 if kv_cache_dtype == "fp8" else 1.0
    scale = head_size**-0.5

    def caches():
        gen = torch.Generator().manual_seed(1)
        if kv_cache_dtype == "fp8":
            kc = torch.randint(0, 120, (num_blocks, kv_heads, head_size // x, block_size, x), generator=gen, dtype=torch.uint8)
            vc = torch.randint(0, 120, (num_blocks, kv_heads, head_size, block_size), generator=gen, dtype=torch.uint8)
        else:
            kc = (torch.rand((num_blocks, kv_heads, head_size // x, block_size, x), generator=gen) - 0.5).to(dtype)
            vc = (torch.rand((num_blocks, kv_heads, head_size, block_size), generator=gen) - 0.5).to(dtype)
        return kc.to(d), vc.to(d)

    pos_d, slots_d, bt_d, sl_d = pos.to(d), slots.to(d), bt.to(d), seq_lens.to(d)
    ref_kc, ref_vc = caches()
    qkv = ops.rotary_embedding_and_cache_partial(pos_d, slab, heads, kv_heads, head_size, cos_sin, ref_kc, ref_vc,
                                                 slots_d, kv_cache_dtype, kv_scale, dtype)
    q = qkv[:, :heads * head_size].reshape(num_seqs, heads, head_size)
    def decode(qq, kcc, vcc):
        # the separate-launch form of what the fused launch takes: it keeps the unpartitioned kernel up to ~900 tokens
        # (PagedAttention.use_v1_fused, measured on MI355X) where forward_decode follows the reference's rule and
        # partitions; v1 and v2 differ in the last bits, so the comparison is form against form
        if PagedAttention.use_v1_fused(max_ctx, num_seqs, heads) and not PagedAttention.use_v1(max_ctx, num_seqs, heads):
            out = torch.empty_like(qq)
            ops.paged_attention_v1(out, qq, kcc, vcc, kv_heads, scale, bt_d, sl_d, block_size, max_ctx, None,
                                   kv_cache_dtype, kv_scale)
            return out
        return PagedAttention.forward_decode(qq, kcc, vcc, bt_d, sl_d, max_ctx, kv_cache_dtype, kv_heads, scale, None,
                                             kv_scale)

    ref = decode(q, ref_kc, ref_vc)
    kc, vc = caches()
    got = PagedAttention.forward_decode_rope_partial(slab, pos_d, cos_sin, slots_d, kc, vc, bt_d, sl_d, max_ctx,
                                                     kv_cache_dtype, heads, kv_heads, head_size, scale, kv_scale, dtype)
    assert torch.equal(kc.view(torch.uint8), ref_kc.view(torch.uint8))
    assert torch.equal(vc.view(torch.uint8), ref_vc.view(torch.uint8))
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16))
    # the same from the finished qkv row in the model dtype (W8A8 / unquantised projections)
    row = slab.sum(0).to(dtype)        # any row will do: the reference below starts from the same bits
    ref_kc2, ref_vc2 = caches()
    rot = row.clone()
    q2, k2, v2 = rot.split([heads * head_size, kv_heads * head_size, kv_heads * head_size], dim=-1)
    ops.rotary_embedding_and_cache(pos_d, q2, k2, v2, head_size, cos_sin, True, ref_kc2, ref_vc2, slots_d,
                                   kv_cache_dtype, kv_scale)
    ref2 = decode(q2.reshape(num_seqs, heads, head_size).contiguous(), ref_kc2, ref_vc2)
    kc2, vc2 = caches()
    got2 = PagedAttention.forward_decode_rope_partial(row, pos_d, cos_sin, slots_d, kc2, vc2, bt_d, sl_d, max_ctx,
                                                      kv_cache_dtype, heads, kv_heads, head_size, scale, kv_scale,
                                                      dtype)
    assert torch.equal(kc2.view(torch.uint8), ref_kc2.view(torch.uint8))
    assert torch.equal(vc2.view(torch.uint8), ref_vc2.view(torch.uint8))
    assert torch.equal(got2.view(torch.int16), ref2.view(torch.int16))


@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("kv_cache_dtype", ["auto", "fp8"])
@pytest.mark.parametrize("num_heads,head_size,block_size,use_alibi", [((32, 8), 128, 16, False), ((40, 40), 80, 32, True),
                                                                      ((8, 2), 256, 8, False)])
def test_paged_attention_float32(gpu_device, version, kv_cache_dtype, num_heads, head_size, block_size, use_alibi):
    """float models (the reference instantiates `float` beside half / bfloat16: attention_kernels.cu:738-766, with a
    float or an fp8 cache): query / output fp32, K cache [.., head / x, block, x] with x = 4 (float) or 16 (fp8);
    against the fp32 torch restatement of the op.  Stated tolerance: 1e-5 abs (fp32 arithmetic in a different
    summation order), 2e-3 with the fp8 cache's scale of 0.5 folded in."""
    inp = helpers.make_paged_attention_inputs(5, 7, num_heads, head_size, block_size, torch.float32,
                                              max_seq_len=1300, num_blocks=256, use_alibi=use_alibi,
                                              kv_cache_dtype=kv_cache_dtype)
    assert inp["key_cache"].shape[-1] == (4 if kv_cache_dtype == "auto" else 16)
    kv_scale = 1.0 if kv_cache_dtype == "auto" else 0.5
    ref = helpers.ref_paged_attention_torch(inp, kv_scale=kv_scale)
    out, parts = run_hip(inp, version, gpu_device, kv_cache_dtype=kv_cache_dtype, kv_scale=kv_scale)
    assert out.dtype == torch.float32
    check(out, ref, atol=2e-5 if kv_cache_dtype == "auto" else 2e-3, rtol=1e-5)
    if parts is not None:
        assert parts[2].dtype == torch.float32


def test_paged_attention_float32_blocksparse(gpu_device):
    inp = helpers.make_paged_attention_inputs(6, 5, (16, 4), 64, 16, torch.float32, max_seq_len=2100, num_blocks=256)
    sparse = dict(tp_rank=0, local_blocks=4, vert_stride=8, block_size=64, head_sliding_step=1)
    ref = helpers.ref_paged_attention_torch(inp, blocksparse=sparse)
    for version in ("v1", "v2"):
        out, _ = run_hip(inp, version, gpu_device, sparse=sparse)
        check(out, ref, atol=2e-5, rtol=1e-5)
