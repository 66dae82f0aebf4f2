"""Logits from the LM head (reference: vllm/model_executor/layers/logits_processor.py:13-113):
plain library GEMM on the vocab shard, gather to TP rank 0, drop the vocabulary padding."""
from typing import Optional

import torch
import torch.nn as nn

from ...distributed import tensor_model_parallel_gather


class LogitsProcessor(nn.Module):

    def __init__(self, vocab_size: int, org_vocab_size: Optional[int] = None, scale: float = 1.0,
                 logits_as_input: bool = False) -> None:
        super().__init__()
        self.scale = scale
        self.vocab_size = vocab_size
        self.logits_as_input = logits_as_input
        self.org_vocab_size = org_vocab_size or vocab_size

    def forward(self, embedding: torch.Tensor, hidden_states: torch.Tensor,
                embedding_bias: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        logits = hidden_states if self.logits_as_input else self._get_logits(
            hidden_states, embedding, embedding_bias)
        if logits is not None and self.scale != 1.0:
            logits *= self.scale
        return logits

    def _get_logits(self, hidden_states, embedding, embedding_bias) -> Optional[torch.Tensor]:
        logits = torch.matmul(hidden_states, embedding.t())
        if embedding_bias is not None:
            logits += embedding_bias
        logits = tensor_model_parallel_gather(logits)
        if logits is not None:
            logits = logits[:, :self.org_vocab_size]
        return logits
