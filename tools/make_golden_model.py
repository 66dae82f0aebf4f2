"""Generate tests/golden/tiny_llama_*.npz / tiny_opt_*.npz from the REFERENCE's own model code run on CPU.
Runs only in the build container (needs /root/reference and oracle/_ref).

What runs (north_star's correctness definition: logits versus the reference CPU executor on the same
inputs): the reference's `LlamaForCausalLM` / `OPTForCausalLM` (vllm/model_executor/models/llama.py,
opt.py) instantiated on CPU in bf16 -- the CPU executor casts fp16 to bf16, cpu_executor.py:114-117 -- with
its TorchSDPA attention backend (vllm/attention/backends/torch_sdpa.py:175-225: torch SDPA for the prompt,
the compiled csrc/cpu paged_attention / reshape_and_cache kernels of oracle/_ref for decode) and its own
layer classes (RMSNorm, rotary embedding, SiluAndMul, the parallel linears in a gloo world of 1).  The
recipe is tests/basic_correctness/test_basic_correctness.py:38-66: a prompt step, then greedy decode.

Weights are generated from CPU seeds by tests/helpers.py (`tiny_llama_checkpoint`), quantised with the
reference's `quantize_weights` for the w4a16 case (the reference model then runs the dequantised `w_ref` as
dense weights -- its CPU executor has no quantised GEMM); the GPU test regenerates the identical tensors
from the same seeds and feeds the GPTQ tensors to the HIP path.  The fixture stores OUTPUTS only: prompt
logits, per-step logits and greedy tokens, the slot mapping used, plus sha256 of the regenerated inputs.

usage:  python tools/make_golden_model.py
"""
import os
import socket
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


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def boot_reference():
    """import the reference with its CPU ops: the kernels of oracle/_ref are registered as _C_ref /
    _C_ref_cache_ops (so that they can live beside this repo's _C ops in the tests); the reference's Python
    looks them up as torch.ops._C / _C_cache_ops"""
    assert build_ref.have_reference(), "/root/reference is not here"
    build_ref.build()
    assert build_ref.load_ref(), "oracle/_ref not loadable"
    torch.ops._C = torch.ops._C_ref
    torch.ops._C_cache_ops = torch.ops._C_ref_cache_ops
    sys.modules.setdefault("cpuinfo", types.ModuleType("cpuinfo"))  # optional dep, absent here
    sys.path.insert(0, build_ref.REF_ROOT)
    import vllm.utils as vu
    # the reference decides "CPU build" from the installed wheel's version string (utils.py is_cpu);
    # it is imported from source here
    vu.is_cpu = lambda: True
    vu.is_hip = lambda: False
    import vllm.attention.selector as sel
    sel.is_cpu, sel.is_hip = vu.is_cpu, vu.is_hip
    import vllm.attention.backends.torch_sdpa  # noqa: F401
    from vllm.distributed import init_distributed_environment, initialize_model_parallel
    init_distributed_environment(1, 0, f"tcp://127.0.0.1:{_free_port()}", 0, backend="gloo")
    initialize_model_parallel(1)


def run_reference(model, kv_shape, prompts, steps, block_size, num_blocks, dtype, vocab):
    """prompt step + `steps` greedy decode steps through the reference model and backend; returns
    (prompt_logits [B, V], step_logits [steps, B, V], tokens [steps + 1, B], block_tables)"""
    from vllm.attention.backends.torch_sdpa import TorchSDPAMetadata
    b, plen = prompts.shape
    nl = kv_shape[0]
    kv_caches = [torch.zeros(kv_shape[1:], dtype=dtype) for _ in range(nl)]
    max_len = plen + steps + 1
    bps = (max_len + block_size - 1) // block_size
    g = torch.Generator().manual_seed(1234)
    block_tables = torch.randperm(num_blocks, generator=g)[:b * bps].to(torch.int32).view(b, bps)

    def slots(pos):  # pos: [B, T] positions -> flat slot mapping
        blk = torch.gather(block_tables.long(), 1, pos // block_size)
        return (blk * block_size + pos % block_size).view(-1)

    pos = torch.arange(plen).repeat(b, 1)
    md = TorchSDPAMetadata(is_prompt=True, slot_mapping=slots(pos), seq_lens=[plen] * b,
                           seq_lens_tensor=torch.full((b, ), plen, dtype=torch.int32), max_decode_seq_len=0,
                           num_prefills=b, num_prefill_tokens=b * plen, num_decode_tokens=0,
                           block_tables=torch.tensor([], dtype=torch.int32))
    with torch.inference_mode():
        hidden = model(prompts.reshape(-1), pos.reshape(-1), kv_caches, md)
        last = hidden.view(b, plen, -1)[:, -1]
        logits = torch.matmul(last, model.lm_head.weight.t() if hasattr(model, "lm_head")
                              else model.lm_head_weight.t())[:, :vocab].float()
        prompt_logits = logits.clone()
        toks = [logits.argmax(-1)]
        step_logits = []
        for s in range(steps):
            p = torch.full((b, 1), plen + s, dtype=torch.int64)
            md = TorchSDPAMetadata(is_prompt=False, slot_mapping=slots(p), seq_lens=[plen + s + 1] * b,
                                   seq_lens_tensor=torch.full((b, ), plen + s + 1, dtype=torch.int32),
                                   max_decode_seq_len=plen + s + 1, num_prefills=0, num_prefill_tokens=0,
                                   num_decode_tokens=b, block_tables=block_tables)
            hidden = model(toks[-1], p.view(-1), kv_caches, md)
            logits = torch.matmul(hidden, model.lm_head.weight.t() if hasattr(model, "lm_head")
                                  else model.lm_head_weight.t())[:, :vocab].float()
            step_logits.append(logits.clone())
            toks.append(logits.argmax(-1))
    return prompt_logits, torch.stack(step_logits), torch.stack(toks), block_tables


def gen_tiny_llama():
    from transformers import LlamaConfig
    from vllm.config import CacheConfig
    from vllm.model_executor.layers.quantization.utils import quant_utils
    from vllm.model_executor.models.llama import LlamaForCausalLM
    a = helpers.TINY_LLAMA
    cfg = LlamaConfig(hidden_size=a["hidden_size"], intermediate_size=a["intermediate_size"],
                      num_hidden_layers=a["num_hidden_layers"], num_attention_heads=a["num_attention_heads"],
                      num_key_value_heads=a["num_key_value_heads"], vocab_size=a["vocab_size"],
                      rms_norm_eps=a["rms_norm_eps"], rope_theta=a["rope_theta"],
                      max_position_embeddings=a["max_position_embeddings"], hidden_act="silu",
                      tie_word_embeddings=False)
    # newer transformers keep rope_theta / rope_scaling inside `rope_parameters`; the 0.5.1 reader wants
    # the flat attributes (llama.py:183-189 reads them with getattr defaults: a missing rope_theta would
    # silently become 10000)
    cfg.rope_scaling = None
    cfg.rope_theta = a["rope_theta"]
    for k in ("hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads",
              "num_key_value_heads", "vocab_size", "rms_norm_eps", "rope_theta", "max_position_embeddings"):
        assert getattr(cfg, k) == a[k], k
    dtype = torch.bfloat16
    hd = a["hidden_size"] // a["num_attention_heads"]
    block_size, num_blocks, b, plen, steps = 16, 64, 3, 21, 6
    for case in ("bf16", "w4a16"):
        ckpt = helpers.tiny_llama_checkpoint(0, dtype)
        dense = {}
        for name, w in ckpt.items():
            if case == "w4a16" and helpers.is_quantised_linear(name):
                # the reference's own quantiser: w_ref is what a CPU executor would run as dense weights
                w_ref, q_w, s, g_idx, _ = quant_utils.quantize_weights(w.t().contiguous(), 4, 128, False)
                oq, os_ = helpers.quantize_like_reference(w.t().contiguous(), 4, 128)
                assert torch.equal(q_w, oq) and torch.equal(s, os_), name   # the test-side quantiser is pinned
                dense[name] = w_ref.t().contiguous().to(dtype)
            else:
                dense[name] = w
        prev = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        try:
            model = LlamaForCausalLM(cfg, cache_config=CacheConfig(block_size, 0.9, 0, "auto"), quant_config=None)
        finally:
            torch.set_default_dtype(prev)
        model.load_weights(iter(dense.items()))
        g = torch.Generator().manual_seed(77)
        prompts = torch.randint(0, a["vocab_size"], (b, plen), generator=g)
        kv_shape = (a["num_hidden_layers"], 2, num_blocks, block_size * a["num_key_value_heads"] * hd)
        pl, sl, toks, bt = run_reference(model, kv_shape, prompts, steps, block_size, num_blocks, dtype,
                                         a["vocab_size"])
        # greedy margins: where the top-2 gap is small a different summation order may flip the token
        top2 = torch.cat([pl[None], sl]).topk(2, dim=-1).values
        path = os.path.join(GOLD, f"tiny_llama_{case}.npz")
        np.savez_compressed(path, case=case, seed=0, prompts=prompts.numpy(), block_tables=bt.numpy(),
                            block_size=block_size, num_blocks=num_blocks, steps=steps,
                            prompt_logits=pl.numpy(), step_logits=sl.numpy(), tokens=toks.numpy(),
                            top2_margin=(top2[..., 0] - top2[..., 1]).numpy(),
                            ckpt_sha=np.array(helpers.tensor_sha(*[ckpt[k] for k in sorted(ckpt)])))
        print(f"  tiny_llama_{case}.npz  {os.path.getsize(path) / 1024:.1f} KiB   tokens {toks.tolist()}")


def gen_tiny_opt():
    """BASELINE.json configs[0]: OPT on the CPU executor, greedy decode -- at OPT-125m's head geometry
    (12 heads x 64, MHA, learned positions, LayerNorm, ReLU, biases) with 2 layers and a small vocabulary"""
    from transformers import OPTConfig
    from vllm.config import CacheConfig
    from vllm.model_executor.models.opt import OPTForCausalLM
    a = helpers.TINY_OPT
    cfg = OPTConfig(hidden_size=a["hidden_size"], ffn_dim=a["ffn_dim"], num_hidden_layers=a["num_hidden_layers"],
                    num_attention_heads=a["num_attention_heads"], vocab_size=a["vocab_size"],
                    max_position_embeddings=a["max_position_embeddings"], word_embed_proj_dim=a["hidden_size"],
                    do_layer_norm_before=True, activation_function="relu", enable_bias=True,
                    layer_norm_elementwise_affine=True)
    dtype = torch.bfloat16
    hd = a["hidden_size"] // a["num_attention_heads"]
    block_size, num_blocks, b, plen, steps = 16, 64, 3, 19, 6
    ckpt = helpers.tiny_opt_checkpoint(0, dtype)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        model = OPTForCausalLM(cfg, cache_config=CacheConfig(block_size, 0.9, 0, "auto"), quant_config=None)
    finally:
        torch.set_default_dtype(prev)
    model.load_weights(iter(ckpt.items()))
    g = torch.Generator().manual_seed(78)
    prompts = torch.randint(0, a["vocab_size"], (b, plen), generator=g)
    kv_shape = (a["num_hidden_layers"], 2, num_blocks, block_size * a["num_attention_heads"] * hd)
    pl, sl, toks, bt = run_reference(model, kv_shape, prompts, steps, block_size, num_blocks, dtype, a["vocab_size"])
    top2 = torch.cat([pl[None], sl]).topk(2, dim=-1).values
    path = os.path.join(GOLD, "tiny_opt_bf16.npz")
    np.savez_compressed(path, seed=0, prompts=prompts.numpy(), block_tables=bt.numpy(), block_size=block_size,
                        num_blocks=num_blocks, steps=steps, prompt_logits=pl.numpy(), step_logits=sl.numpy(),
                        tokens=toks.numpy(), top2_margin=(top2[..., 0] - top2[..., 1]).numpy(),
                        ckpt_sha=np.array(helpers.tensor_sha(*[ckpt[k] for k in sorted(ckpt)])))
    print(f"  tiny_opt_bf16.npz  {os.path.getsize(path) / 1024:.1f} KiB   tokens {toks.tolist()}")


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    boot_reference()
    gen_tiny_llama()
    gen_tiny_opt()
    print("done")
