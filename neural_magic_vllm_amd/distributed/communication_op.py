"""vllm/distributed/communication_op.py:1-33: thin wrappers over the TP group."""
from typing import Any, Dict, Optional, Union

import torch

from .parallel_state import get_tp_group, model_parallel_is_initialized


def tensor_model_parallel_all_reduce(input_: torch.Tensor) -> torch.Tensor:
    """All-reduce the input tensor across the tensor-parallel group (RCCL over xGMI)."""
    if not model_parallel_is_initialized():
        return input_
    return get_tp_group().all_reduce(input_)


def tensor_model_parallel_all_gather(input_: torch.Tensor, dim: int = -1) -> torch.Tensor:
    if not model_parallel_is_initialized():
        return input_
    return get_tp_group().all_gather(input_, dim)


def tensor_model_parallel_gather(input_: torch.Tensor, dst: int = 0,
                                 dim: int = -1) -> Optional[torch.Tensor]:
    if not model_parallel_is_initialized():
        return input_
    return get_tp_group().gather(input_, dst, dim)


def broadcast_tensor_dict(tensor_dict: Optional[Dict[Any, Union[torch.Tensor, Any]]] = None,
                          src: int = 0):
    if not model_parallel_is_initialized():
        return tensor_dict
    return get_tp_group().broadcast_tensor_dict(tensor_dict, src)
