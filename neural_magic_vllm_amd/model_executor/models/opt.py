"""OPT decoder on the gfx950 hot path -- BASELINE.json configs[0] (OPT-125m greedy decode, the reference's
own CPU-runnable case) needs a second model family to drive the kernels through: multi-head attention
with head size 64 (paged attention, cache write and prompt attention at a geometry Llama does not have),
biased projections, LayerNorm and ReLU.

Structure follows vllm/model_executor/models/opt.py of the reference: OPTLearnedPositionalEmbedding
(:45-57, position ids offset by 2), OPTAttention (:60-112), OPTDecoderLayer (:115-188), OPTDecoder
(:191-260), OPTForCausalLM (:283-357, lm_head tied to the token embedding).  LayerNorm and ReLU are torch's
own, as in the reference; `word_embed_proj_dim != hidden_size` (project_in / project_out, OPT-350m only)
and post-layer-norm checkpoints are not built.
"""
from typing import Any, Iterable, List, Optional, Tuple

import torch
from torch import nn

from ...attention import Attention, AttentionMetadata
from ...distributed import get_tensor_model_parallel_world_size
from ..layers.linear import ColumnParallelLinear, QKVParallelLinear, RowParallelLinear
from ..layers.logits_processor import LogitsProcessor
from ..layers.quantization.base_config import QuantizationConfig
from ..layers.vocab_parallel_embedding import VocabParallelEmbedding


class OPTLearnedPositionalEmbedding(nn.Embedding):
    """learned absolute positions; OPT keeps two padding rows in front of the table (opt.py:45-57)"""
    OFFSET = 2

    def __init__(self, num_embeddings: int, embedding_dim: int):
        super().__init__(num_embeddings + self.OFFSET, embedding_dim)

    def forward(self, positions: torch.Tensor):
        return super().forward(positions + self.OFFSET)


class OPTAttention(nn.Module):

    def __init__(self, embed_dim: int, num_heads: int, bias: bool = True, cache_config: Optional[Any] = None,
                 quant_config: Optional[QuantizationConfig] = None) -> None:
        super().__init__()
        tp = get_tensor_model_parallel_world_size()
        assert num_heads % tp == 0
        self.num_heads = num_heads // tp
        self.head_dim = embed_dim // num_heads
        self.qkv_proj = QKVParallelLinear(embed_dim, self.head_dim, num_heads, bias=bias, quant_config=quant_config)
        self.out_proj = RowParallelLinear(embed_dim, embed_dim, bias=bias, quant_config=quant_config)
        self.attn = Attention(self.num_heads, self.head_dim, self.head_dim**-0.5, cache_config=cache_config,
                              quant_config=quant_config)

    def forward(self, hidden_states: torch.Tensor, kv_cache: Optional[torch.Tensor],
                attn_metadata: AttentionMetadata) -> torch.Tensor:
        qkv, _ = self.qkv_proj(hidden_states)
        q, k, v = qkv.chunk(3, dim=-1)
        out, _ = self.out_proj(self.attn(q, k, v, kv_cache, attn_metadata))
        return out


class OPTDecoderLayer(nn.Module):
    """pre-LayerNorm block (do_layer_norm_before, every OPT but 350m): x += attn(ln1(x)); x += fc2(relu(fc1(ln2(x))))"""

    def __init__(self, config, cache_config: Optional[Any] = None,
                 quant_config: Optional[QuantizationConfig] = None) -> None:
        super().__init__()
        if not getattr(config, "do_layer_norm_before", True):
            raise ValueError("post-layer-norm OPT checkpoints (opt-350m) are not supported")
        if getattr(config, "activation_function", "relu") != "relu":
            raise ValueError(f"Unsupported activation: {config.activation_function}. Only relu is supported.")
        h = config.hidden_size
        bias = getattr(config, "enable_bias", True)
        affine = getattr(config, "layer_norm_elementwise_affine", True)
        self.self_attn = OPTAttention(h, config.num_attention_heads, bias, cache_config, quant_config)
        self.self_attn_layer_norm = nn.LayerNorm(h, elementwise_affine=affine)
        self.fc1 = ColumnParallelLinear(h, config.ffn_dim, bias=bias, quant_config=quant_config)
        self.fc2 = RowParallelLinear(config.ffn_dim, h, bias=bias, quant_config=quant_config)
        self.final_layer_norm = nn.LayerNorm(h, elementwise_affine=affine)

    def forward(self, hidden_states: torch.Tensor, kv_cache: Optional[torch.Tensor],
                attn_metadata: AttentionMetadata) -> torch.Tensor:
        hidden_states = hidden_states + self.self_attn(self.self_attn_layer_norm(hidden_states), kv_cache,
                                                       attn_metadata)
        x, _ = self.fc1(self.final_layer_norm(hidden_states))
        x, _ = self.fc2(torch.relu(x))
        return hidden_states + x


class OPTDecoder(nn.Module):

    def __init__(self, config, cache_config: Optional[Any] = None,
                 quant_config: Optional[QuantizationConfig] = None) -> None:
        super().__init__()
        if getattr(config, "word_embed_proj_dim", config.hidden_size) != config.hidden_size:
            raise ValueError("word_embed_proj_dim != hidden_size (opt-350m) is not supported")
        self.embed_tokens = VocabParallelEmbedding(config.vocab_size, config.hidden_size)
        self.embed_positions = OPTLearnedPositionalEmbedding(config.max_position_embeddings, config.hidden_size)
        self.layers = nn.ModuleList([OPTDecoderLayer(config, cache_config, quant_config)
                                     for _ in range(config.num_hidden_layers)])
        self.final_layer_norm = nn.LayerNorm(config.hidden_size,
                                             elementwise_affine=getattr(config, "layer_norm_elementwise_affine", True))

    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor, kv_caches: List[Optional[torch.Tensor]],
                attn_metadata: AttentionMetadata) -> torch.Tensor:
        x = self.embed_tokens(input_ids) + self.embed_positions(positions)
        for layer, kv in zip(self.layers, kv_caches):
            x = layer(x, kv, attn_metadata)
        return self.final_layer_norm(x)


class OPTModel(nn.Module):

    def __init__(self, config, cache_config=None, quant_config=None) -> None:
        super().__init__()
        self.decoder = OPTDecoder(config, cache_config, quant_config)

    def forward(self, input_ids, positions, kv_caches, attn_metadata):
        return self.decoder(input_ids, positions, kv_caches, attn_metadata)


class OPTForCausalLM(nn.Module):

    def __init__(self, config, cache_config: Optional[Any] = None,
                 quant_config: Optional[QuantizationConfig] = None) -> None:
        super().__init__()
        self.config = config
        self.model = OPTModel(config, cache_config, quant_config)
        self.lm_head = self.model.decoder.embed_tokens       # tied (opt.py:296)
        self.logits_processor = LogitsProcessor(config.vocab_size)

    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor, kv_caches: List[Optional[torch.Tensor]],
                attn_metadata: AttentionMetadata) -> torch.Tensor:
        return self.model(input_ids, positions, kv_caches, attn_metadata)

    def compute_logits(self, hidden_states: torch.Tensor) -> Optional[torch.Tensor]:
        return self.logits_processor(self.lm_head.weight, hidden_states)

    def load_weights(self, weights: Iterable[Tuple[str, torch.Tensor]]) -> None:
        """HF names (`model.decoder.*` or `decoder.*`); q/k/v are stacked into qkv_proj (opt.py:325-357)"""
        params = dict(self.named_parameters(remove_duplicate=False))
        loaded = set()
        for name, w in weights:
            if "lm_head.weight" in name:
                continue
            if name.startswith("decoder."):
                name = "model." + name
            for shard, tag in (("q_proj", "q"), ("k_proj", "k"), ("v_proj", "v")):
                if f".{shard}." in name:
                    pname = name.replace(shard, "qkv_proj")
                    params[pname].weight_loader(params[pname], w, tag)
                    loaded.add(pname)
                    break
            else:
                prm = params[name]
                loader = getattr(prm, "weight_loader", None)
                if loader is not None:
                    loader(prm, w)
                else:
                    assert prm.shape == w.shape, name
                    prm.data.copy_(w)
                loaded.add(name)
        missing = sorted(n for n in params if n not in loaded and not n.startswith("lm_head"))
        if missing:
            raise ValueError(f"checkpoint has no tensor for: {missing[:8]}{' ...' if len(missing) > 8 else ''}")
