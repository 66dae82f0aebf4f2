"""Vocab-parallel embedding and LM head (reference: vllm/model_executor/layers/
vocab_parallel_embedding.py:130-420, without the LoRA added-vocabulary bookkeeping)."""
from typing import Optional

import torch
import torch.nn.functional as F
from torch.nn.parameter import Parameter

from ...distributed import (get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size,
                            tensor_model_parallel_all_reduce)
from ..utils import set_weight_attrs

DEFAULT_VOCAB_PADDING_SIZE = 64


def pad_vocab_size(vocab_size: int, pad_to: int = DEFAULT_VOCAB_PADDING_SIZE) -> int:
    return ((vocab_size + pad_to - 1) // pad_to) * pad_to


class VocabParallelEmbedding(torch.nn.Module):
    """Embedding sharded along the vocabulary; out-of-shard ids contribute zeros and the
    partial lookups are summed with one all-reduce (vocab_parallel_embedding.py:333-351)."""

    def __init__(self, num_embeddings: int, embedding_dim: int,
                 params_dtype: Optional[torch.dtype] = None,
                 org_num_embeddings: Optional[int] = None,
                 padding_size: int = DEFAULT_VOCAB_PADDING_SIZE):
        super().__init__()
        self.num_embeddings = num_embeddings
        self.org_vocab_size = org_num_embeddings or num_embeddings
        self.num_embeddings_padded = pad_vocab_size(num_embeddings, padding_size)
        self.embedding_dim = embedding_dim
        if params_dtype is None:
            params_dtype = torch.get_default_dtype()
        self.tp_size = get_tensor_model_parallel_world_size()
        tp_rank = get_tensor_model_parallel_rank()
        assert self.num_embeddings_padded % self.tp_size == 0
        self.num_embeddings_per_partition = self.num_embeddings_padded // self.tp_size
        self.vocab_start_index = tp_rank * self.num_embeddings_per_partition
        self.vocab_end_index = self.vocab_start_index + self.num_embeddings_per_partition
        self.weight = Parameter(torch.empty(self.num_embeddings_per_partition, self.embedding_dim,
                                            dtype=params_dtype))
        set_weight_attrs(self.weight, {"parallel_dim": 0, "weight_loader": self.weight_loader})

    def weight_loader(self, param: Parameter, loaded_weight: torch.Tensor):
        assert loaded_weight.shape[0] == self.org_vocab_size
        start = self.vocab_start_index
        end = min(self.vocab_end_index, self.org_vocab_size)
        n = max(end - start, 0)
        param.data[:n].copy_(loaded_weight[start:start + n])
        param.data[n:].fill_(0)

    def forward(self, input_):
        if self.tp_size > 1:
            mask = (input_ < self.vocab_start_index) | (input_ >= self.vocab_end_index)
            masked_input = input_.clone() - self.vocab_start_index
            masked_input[mask] = 0
        else:
            masked_input = input_
        output_parallel = F.embedding(masked_input, self.weight)
        if self.tp_size > 1:
            output_parallel[mask, :] = 0.0
        return tensor_model_parallel_all_reduce(output_parallel)


class ParallelLMHead(VocabParallelEmbedding):
    """Output head; logits are computed by LogitsProcessor from .weight"""

    def __init__(self, num_embeddings: int, embedding_dim: int, bias: bool = False,
                 params_dtype: Optional[torch.dtype] = None,
                 org_num_embeddings: Optional[int] = None,
                 padding_size: int = DEFAULT_VOCAB_PADDING_SIZE):
        super().__init__(num_embeddings, embedding_dim, params_dtype, org_num_embeddings,
                         padding_size)
        if bias:
            self.bias = Parameter(torch.empty(self.num_embeddings_per_partition,
                                              dtype=params_dtype))
            set_weight_attrs(self.bias, {"parallel_dim": 0, "weight_loader": self.weight_loader})
        else:
            self.register_parameter("bias", None)

    def forward(self, input_):
        del input_
        raise RuntimeError("LMHead's weights should be used in the sampler.")
