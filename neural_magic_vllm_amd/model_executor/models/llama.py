"""Llama decoder on the gfx950 hot path -- the harness that drives the kernels end to end.

Structure follows vllm/model_executor/models/llama.py of the reference: LlamaMLP (:51-85),
LlamaAttention (:88-166), LlamaDecoderLayer (:169-241), LlamaModel (:244-321),
LlamaForCausalLM (:324-515): fused qkv / gate_up projections through the LinearMethod plugin
surface, rope in place on the q/k slices, attention through the Attention layer, residual
carried through fused_add_rms_norm.  Checkpoint name mapping, LoRA, pipeline parallelism and
rope scaling are outside the hot-path scope.
"""
import os
import warnings
from typing import Any, Iterable, List, NamedTuple, Optional, Tuple

import torch
from torch import nn

from ... import _custom_ops as ops
from ...attention import Attention, AttentionMetadata
from ...distributed import get_tensor_model_parallel_world_size, get_tp_group
from ..layers.activation import SiluAndMul
from ..layers.layernorm import RMSNorm
from ..layers.linear import MergedColumnParallelLinear, QKVParallelLinear, RowParallelLinear
from ..layers.logits_processor import LogitsProcessor
from ..layers.quantization.base_config import QuantizationConfig
from ..layers.quantization.compressed_tensors import Int8Activations, accepts_int8_activations
from ..layers.rotary_embedding import get_rope
from ..layers.vocab_parallel_embedding import ParallelLMHead, VocabParallelEmbedding


def fused_glue_default() -> bool:
    """Fused decode-step launches (rotary + cache write; norm / activation + int8 quantisation) are
    on unless NMV_FUSED_GLUE=0; every one is bit-identical to the op sequence it replaces.  Modules
    keep the choice in `.fused_glue` so that tests can compare both forms on one set of weights."""
    return os.environ.get("NMV_FUSED_GLUE", "1") != "0"


class SplitKPartial(NamedTuple):
    """the un-reduced output of a row-parallel projection: fp32 split-K slabs [splits, T, hidden]
    (RowParallelLinear.forward_partial); the next fused_add_rms_norm sums them"""
    slab: torch.Tensor


def _add_norm(norm: RMSNorm, hidden_states, residual: torch.Tensor):
    """fused_add_rms_norm whose input may be a SplitKPartial; returns (normed, residual)"""
    if isinstance(hidden_states, SplitKPartial):
        if get_tensor_model_parallel_world_size() > 1:
            # row-parallel projection under TP: all-reduce + residual-add + RMSNorm in one launch
            return get_tp_group().custom_ar.all_reduce_add_rms_norm(
                hidden_states.slab, residual, norm.weight.data, norm.variance_epsilon), residual
        return ops.fused_add_rms_norm_partial(hidden_states.slab, residual, norm.weight.data,
                                              norm.variance_epsilon), residual
    return norm(hidden_states, residual)


_warned = set()


def _warn_once(msg: str) -> None:
    if msg.split("(e.g.")[0] not in _warned:
        _warned.add(msg.split("(e.g.")[0])
        warnings.warn(msg, stacklevel=3)


class LlamaMLP(nn.Module):

    def __init__(self, hidden_size: int, intermediate_size: int, hidden_act: str,
                 quant_config: Optional[QuantizationConfig] = None, bias: bool = False) -> None:
        super().__init__()
        self.gate_up_proj = MergedColumnParallelLinear(input_size=hidden_size,
                                                       output_sizes=[intermediate_size] * 2,
                                                       bias=bias, quant_config=quant_config)
        self.down_proj = RowParallelLinear(input_size=intermediate_size, output_size=hidden_size,
                                           bias=bias, quant_config=quant_config)
        if hidden_act != "silu":
            raise ValueError(f"Unsupported activation: {hidden_act}. Only silu is supported for now.")
        self.act_fn = SiluAndMul()
        self.fused_glue = fused_glue_default()

    def forward(self, x):
        qm = self.gate_up_proj.quant_method
        if self.fused_glue and not isinstance(x, Int8Activations) and x.is_cuda \
                and hasattr(qm, "apply_silu_mul") and qm.can_fuse_silu_mul(self.gate_up_proj):
            # W4A16: silu_and_mul folded into the gate_up GEMM's epilogue (column-interleaved weights)
            x = qm.apply_silu_mul(self.gate_up_proj, x)
            return self._down(x)
        gate_up, _ = self.gate_up_proj(x)
        if self.fused_glue and gate_up.is_cuda and accepts_int8_activations(self.down_proj):
            # silu_and_mul + dynamic per-token int8 quantisation of down_proj's input: one launch
            x = Int8Activations(*ops.silu_and_mul_dynamic_int8_quant(gate_up), gate_up.dtype)
        else:
            x = self.act_fn(gate_up)
        return self._down(x)

    def _down(self, x):
        if self.fused_glue and not isinstance(x, Int8Activations):
            slab = self.down_proj.forward_partial(x, allow16=True)
            if slab is not None:
                return SplitKPartial(slab)   # summed by the next layer's input_layernorm
        x, _ = self.down_proj(x)
        return x


class LlamaAttention(nn.Module):

    def __init__(self, config, hidden_size: int, num_heads: int, num_kv_heads: int,
                 rope_theta: float = 10000, rope_scaling: Optional[dict] = None,
                 max_position_embeddings: int = 8192,
                 quant_config: Optional[QuantizationConfig] = None, bias: bool = False,
                 cache_config: Optional[Any] = None) -> None:
        super().__init__()
        self.hidden_size = hidden_size
        tp_size = get_tensor_model_parallel_world_size()
        self.total_num_heads = num_heads
        assert self.total_num_heads % tp_size == 0
        self.num_heads = self.total_num_heads // tp_size
        self.total_num_kv_heads = num_kv_heads
        if self.total_num_kv_heads >= tp_size:
            assert self.total_num_kv_heads % tp_size == 0
        else:
            # fewer KV heads than ranks: replicate them (llama.py:109-117)
            assert tp_size % self.total_num_kv_heads == 0
        self.num_kv_heads = max(1, self.total_num_kv_heads // tp_size)
        self.head_dim = getattr(config, "head_dim", None) or self.hidden_size // self.total_num_heads
        self.q_size = self.num_heads * self.head_dim
        self.kv_size = self.num_kv_heads * self.head_dim
        self.scaling = self.head_dim**-0.5
        self.qkv_proj = QKVParallelLinear(hidden_size=hidden_size, head_size=self.head_dim,
                                          total_num_heads=self.total_num_heads,
                                          total_num_kv_heads=self.total_num_kv_heads, bias=bias,
                                          quant_config=quant_config)
        self.o_proj = RowParallelLinear(input_size=self.total_num_heads * self.head_dim,
                                        output_size=hidden_size, bias=bias,
                                        quant_config=quant_config)
        self.rotary_emb = get_rope(self.head_dim, rotary_dim=self.head_dim,
                                   max_position=max_position_embeddings, base=rope_theta,
                                   rope_scaling=rope_scaling)
        self.attn = Attention(self.num_heads, self.head_dim, self.scaling,
                              num_kv_heads=self.num_kv_heads, cache_config=cache_config,
                              quant_config=quant_config)
        self.fused_glue = fused_glue_default()
        # experiment switch: "1" = rope + cache write in the attention launch's prologue (default), "0" = in their own
        # launch between the qkv GEMM and the attention
        self.attn_prologue = os.environ.get("NMV_ATTN_PROLOGUE", "1")

    def forward(self, positions: torch.Tensor, hidden_states: torch.Tensor,
                kv_cache: Optional[torch.Tensor], attn_metadata: AttentionMetadata) -> torch.Tensor:
        qkv = None
        if self.fused_glue and isinstance(hidden_states, torch.Tensor) and hidden_states.dim() == 2 \
                and self.attn_prologue != "plain":
            # deferred split-K: the rope + cache launch sums the qkv projection's fp32 slabs
            # a prompt step's slabs go to the rope + cache launch, which also reads them in the model dtype; a decode
            # step's go to the attention launch's prologue (fp32 only)
            prompt_only = getattr(attn_metadata, "num_prefill_tokens", 0) > 0 and getattr(attn_metadata, "num_decode_tokens", 0) == 0
            slab = self.qkv_proj.forward_partial(hidden_states, allow16=prompt_only)
            if slab is not None:
                # decode-only batch: rope + cache write + paged attention in one launch
                attn_output = None
                if self.attn_prologue == "1":
                    attn_output = self.attn.decode_rope_partial(positions, slab, self.rotary_emb, kv_cache,
                                                                attn_metadata, hidden_states.dtype)
                if attn_output is not None:
                    return self._o(attn_output)
                qkv = self.attn.rope_and_cache_partial(positions, slab, self.rotary_emb, kv_cache,
                                                       attn_metadata, hidden_states.dtype)
        if qkv is not None:
            q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)
            attn_output = self.attn(q, k, v, kv_cache, attn_metadata, cache_written=True)
            return self._o(attn_output)
        qkv, _ = self.qkv_proj(hidden_states)
        if self.fused_glue and qkv.is_cuda and qkv.dim() == 2 and self.attn_prologue == "1":
            # decode-only batch: rope + cache write + paged attention in one launch, from the finished row
            attn_output = self.attn.decode_rope_partial(positions, qkv, self.rotary_emb, kv_cache,
                                                        attn_metadata, qkv.dtype)
            if attn_output is not None:
                return self._o(attn_output)
        q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)
        if self.fused_glue and q.is_cuda and self.attn.rope_and_cache(positions, q, k, v, self.rotary_emb,
                                                                      kv_cache, attn_metadata):
            attn_output = self.attn(q, k, v, kv_cache, attn_metadata, cache_written=True)
        else:
            q, k = self.rotary_emb(positions, q, k)
            attn_output = self.attn(q, k, v, kv_cache, attn_metadata)
        return self._o(attn_output)

    def _o(self, attn_output: torch.Tensor):
        if self.fused_glue:
            slab = self.o_proj.forward_partial(attn_output, allow16=True)
            if slab is not None:
                return SplitKPartial(slab)   # summed by post_attention_layernorm
        output, _ = self.o_proj(attn_output)
        return output


class LlamaDecoderLayer(nn.Module):

    def __init__(self, config, cache_config: Optional[Any] = None,
                 quant_config: Optional[QuantizationConfig] = None) -> None:
        super().__init__()
        self.hidden_size = config.hidden_size
        rope_theta = getattr(config, "rope_theta", 10000)
        rope_scaling = getattr(config, "rope_scaling", None)
        max_position_embeddings = getattr(config, "max_position_embeddings", 8192)
        attention_bias = getattr(config, "attention_bias", False) or getattr(config, "bias", False)
        self.self_attn = LlamaAttention(
            config=config, hidden_size=self.hidden_size, num_heads=config.num_attention_heads,
            num_kv_heads=getattr(config, "num_key_value_heads", config.num_attention_heads),
            rope_theta=rope_theta, rope_scaling=rope_scaling,
            max_position_embeddings=max_position_embeddings, quant_config=quant_config,
            bias=attention_bias, cache_config=cache_config)
        self.mlp = LlamaMLP(hidden_size=self.hidden_size,
                            intermediate_size=config.intermediate_size,
                            hidden_act=config.hidden_act, quant_config=quant_config,
                            bias=getattr(config, "mlp_bias", False))
        self.input_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.fused_glue = fused_glue_default()

    def forward(self, positions: torch.Tensor, hidden_states: torch.Tensor,
                kv_cache: Optional[torch.Tensor], attn_metadata: AttentionMetadata,
                residual: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
        fuse_q = self.fused_glue and isinstance(hidden_states, torch.Tensor) and hidden_states.is_cuda
        if fuse_q and accepts_int8_activations(self.self_attn.qkv_proj):
            # (fused_add_)rms_norm + dynamic per-token int8 quantisation of qkv_proj's input
            ln = self.input_layernorm
            if residual is None:
                residual = hidden_states
                q, sc = ops.rms_norm_dynamic_int8_quant(hidden_states, None, ln.weight.data, ln.variance_epsilon)
            else:
                q, sc = ops.rms_norm_dynamic_int8_quant(hidden_states, residual, ln.weight.data,
                                                        ln.variance_epsilon)
            hidden_states = Int8Activations(q, sc, residual.dtype)
        elif residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = _add_norm(self.input_layernorm, hidden_states, residual)
        hidden_states = self.self_attn(positions=positions, hidden_states=hidden_states,
                                       kv_cache=kv_cache, attn_metadata=attn_metadata)
        if fuse_q and accepts_int8_activations(self.mlp.gate_up_proj):
            ln = self.post_attention_layernorm
            q, sc = ops.rms_norm_dynamic_int8_quant(hidden_states, residual, ln.weight.data,
                                                    ln.variance_epsilon)
            hidden_states = Int8Activations(q, sc, residual.dtype)
        else:
            hidden_states, residual = _add_norm(self.post_attention_layernorm, hidden_states, residual)
        hidden_states = self.mlp(hidden_states)
        return hidden_states, residual


class LlamaModel(nn.Module):

    def __init__(self, config, cache_config: Optional[Any] = None,
                 quant_config: Optional[QuantizationConfig] = None) -> None:
        super().__init__()
        self.config = config
        self.vocab_size = config.vocab_size
        self.embed_tokens = VocabParallelEmbedding(self.vocab_size, config.hidden_size,
                                                   org_num_embeddings=config.vocab_size)
        self.layers = nn.ModuleList([LlamaDecoderLayer(config, cache_config, quant_config)
                                     for _ in range(config.num_hidden_layers)])
        self.norm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def forward(self, input_ids: Optional[torch.Tensor], positions: torch.Tensor,
                kv_caches: List[Optional[torch.Tensor]], attn_metadata: AttentionMetadata,
                inputs_embeds: Optional[torch.Tensor] = None) -> torch.Tensor:
        hidden_states = inputs_embeds if inputs_embeds is not None else self.embed_tokens(input_ids)
        residual = None
        side = self._prefetch_stream(hidden_states)
        main = torch.cuda.current_stream(hidden_states.device) if side is not None else None
        for i, layer in enumerate(self.layers):
            if side is not None and i + 1 < len(self.layers):
                # MI355X: while layer i computes, a side stream reads layer i + 1's weights into the Infinity Cache once
                # (ops.prefetch_l3: loads only) -- under graph capture this is a parallel branch of the step's graph
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    for t in self._layer_weight_tensors(i + 1):
                        ops.prefetch_l3(t, self.prefetch_workgroups)
            hidden_states, residual = layer(positions, hidden_states, kv_caches[i], attn_metadata,
                                            residual)
        if side is not None:
            main.wait_stream(side)
        hidden_states, _ = _add_norm(self.norm, hidden_states, residual)
        return hidden_states

    # ---- Infinity-Cache prefetch of the next layer's weights (small decode batches) ---------------------------------------
    prefetch_max_rows = int(os.environ.get("NMV_PREFETCH_MAX_ROWS", "16"))
    prefetch_workgroups = int(os.environ.get("NMV_PREFETCH_WGS", "64"))

    def _prefetch_stream(self, hidden_states: torch.Tensor):
        """the side stream, or None when the hint is off (the default: NMV_PREFETCH=1 turns it on), for a CPU tensor, or
        for more rows than a layer leaves HBM idle for.  Measured on MI355X, round 4 (tools/debug/ab_prefetch.sh,
        profiles/r04_prefetch_branch.txt): the GEMMs alone gain 38.3 -> 31.5 us per layer at M = 1 when their weights are
        cache-resident, but as a second branch of the step's hipGraph the load-only launches cost more than that: the
        runtime splits a forked graph into segments on several queues with 50-100 us gaps between them (B = 1: 2.10 ->
        3.4-3.6 ms per step at 64 / 256 / 1024 workgroups).  Off until the prefetch can ride inside the step's own launches."""
        if not hidden_states.is_cuda or hidden_states.shape[0] > self.prefetch_max_rows:
            return None
        if os.environ.get("NMV_PREFETCH", "0") != "1":
            return None
        s = getattr(self, "_side_stream", None)
        if s is None or s.device != hidden_states.device:
            s = self._side_stream = torch.cuda.Stream(device=hidden_states.device)
        return s

    def _layer_weight_tensors(self, i: int):
        """the tensors layer i's four projections read at decode sizes: the MFMA-native copy where a layer has one, else its
        weight tensor as loaded (scales are a percent of the bytes and ride along)"""
        cache = self.__dict__.setdefault("_prefetch_lists", {})
        got = cache.get(i)
        if got is None:
            layer, got, complete = self.layers[i], [], True
            for lin in (layer.self_attn.qkv_proj, layer.self_attn.o_proj, layer.mlp.gate_up_proj, layer.mlp.down_proj):
                if getattr(lin, "qweight_native", None) is not None:
                    got += [lin.qweight_native, lin.scales_native]
                    continue
                complete = complete and not hasattr(lin, "qweight")   # a native copy may still be built at first use
                for name in ("qweight", "weight", "scales", "weight_scale"):
                    t = getattr(lin, name, None)
                    if isinstance(t, torch.Tensor) and t.is_cuda and t.is_contiguous() and t.numel() * t.element_size() >= 4096:
                        got.append(t.data)
            if complete:
                cache[i] = got
        return got


class LlamaForCausalLM(nn.Module):

    def __init__(self, config, cache_config: Optional[Any] = None,
                 quant_config: Optional[QuantizationConfig] = None) -> None:
        super().__init__()
        self.config = config
        self.model = LlamaModel(config, cache_config, quant_config)
        self.unpadded_vocab_size = config.vocab_size
        # bf16 unless quant_config.lm_head_quantized (gptq_marlin.py:151-157)
        self.lm_head = ParallelLMHead(self.unpadded_vocab_size, config.hidden_size,
                                      org_num_embeddings=config.vocab_size)
        self.tie_word_embeddings = bool(getattr(config, "tie_word_embeddings", False))
        if self.tie_word_embeddings:
            self.lm_head.weight = self.model.embed_tokens.weight   # llama.py:385-386
        self.logits_processor = LogitsProcessor(self.unpadded_vocab_size, config.vocab_size,
                                                getattr(config, "logit_scale", 1.0))

    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor,
                kv_caches: List[Optional[torch.Tensor]],
                attn_metadata: AttentionMetadata) -> torch.Tensor:
        return self.model(input_ids, positions, kv_caches, attn_metadata)

    def compute_logits(self, hidden_states: torch.Tensor) -> Optional[torch.Tensor]:
        return self.logits_processor(self.lm_head.weight, hidden_states)

    def load_weights(self, weights: Iterable[Tuple[str, torch.Tensor]]):
        """stacked-parameter mapping of the reference (llama.py:391-460): q/k/v -> qkv_proj,
        gate/up -> gate_up_proj; `.kv_scale` -> `.attn.kv_scale` (:470-481); everything else by name.
        Unlike the reference this loader is strict: a parameter that no tensor of the checkpoint wrote
        raises (a tied-embedding checkpoint without lm_head.weight is served by the tie below, :385-386)."""
        stacked = [(".qkv_proj", ".q_proj", "q"), (".qkv_proj", ".k_proj", "k"),
                   (".qkv_proj", ".v_proj", "v"), (".gate_up_proj", ".gate_proj", 0),
                   (".gate_up_proj", ".up_proj", 1)]
        params = dict(self.named_parameters())
        written = {}
        for name, loaded in weights:
            if "rotary_emb.inv_freq" in name or "rotary_emb.cos_cached" in name or "rotary_emb.sin_cached" in name:
                continue   # buffers some exporters serialise (llama.py:431-437 skips all three)
            if self.tie_word_embeddings and name == "lm_head.weight":
                continue   # tied: the embedding's tensor is the head's
            for pname, wname, shard_id in stacked:
                if wname not in name:
                    continue
                name = name.replace(wname, pname)
                if name.endswith(".bias") and name not in params:
                    break  # extra bias of GPTQ exports (llama.py:453-455)
                if name not in params:
                    raise ValueError(f"checkpoint tensor {name} has no parameter in the model")
                param = params[name]
                param.weight_loader(param, loaded, shard_id)
                written.setdefault(name, set()).add(shard_id)
                break
            else:
                if name.endswith(".bias") and name not in params:
                    continue
                if name.endswith("kv_scale"):
                    # fp8 checkpoints: the KV-cache scaling factor lives on the Attention layer
                    remapped = name.replace(".kv_scale", ".attn.kv_scale")
                    if remapped not in params:
                        # llama.py:470-481: warn once and carry on with a scale of 1.0
                        _warn_once(f"Found kv scale in the checkpoint (e.g. {name}), but not found the expected name "
                                   f"in the model (e.g. {remapped}). kv-scale is not loaded.")
                        continue
                    name = remapped
                if name not in params:
                    raise ValueError(f"checkpoint tensor {name} has no parameter in the model")
                param = params[name]
                loader = getattr(param, "weight_loader", None)
                if loader is not None:
                    loader(param, loaded)
                else:
                    assert param.shape == loaded.shape, name
                    param.data.copy_(loaded)
                written.setdefault(name, set()).add(None)
        need = {".qkv_proj": {"q", "k", "v"}, ".gate_up_proj": {0, 1}}
        missing = []
        for name, prm in params.items():
            if prm.device.type == "meta" or (self.tie_word_embeddings and name == "lm_head.weight"):
                continue
            if name.endswith(("g_idx", "g_idx_sort_indices", "workspace", "qzeros", "kv_scale", "input_scale")):
                continue   # optional in checkpoints / created by the method itself
            want = next((v for k, v in need.items() if k in name), {None})
            if not want <= written.get(name, set()):
                missing.append(name)
        if missing:
            raise ValueError(f"checkpoint left {len(missing)} parameter(s) unwritten: {missing[:6]}"
                             f"{' ...' if len(missing) > 6 else ''}")
