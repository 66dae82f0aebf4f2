"""The collectives model code calls (interface: reference vllm/distributed/communication_op.py:1-33): each forwards to
the tensor-parallel group's method of the same meaning -- RCCL over xGMI, or the P2P path of ../custom_all_reduce.py --
and is the identity when no group exists (a single-GPU run never initialises one)."""
from typing import Any, Dict, Optional, Union

import torch

from . import parallel_state as _ps


def _group_or_none():
    return _ps.get_tp_group() if _ps.model_parallel_is_initialized() else None


def tensor_model_parallel_all_reduce(input_: torch.Tensor) -> torch.Tensor:
    group = _group_or_none()
    return input_ if group is None else group.all_reduce(input_)


def tensor_model_parallel_all_gather(input_: torch.Tensor, dim: int = -1) -> torch.Tensor:
    group = _group_or_none()
    return input_ if group is None else group.all_gather(input_, dim)


def tensor_model_parallel_gather(input_: torch.Tensor, dst: int = 0, dim: int = -1) -> Optional[torch.Tensor]:
    """rank `dst` receives the concatenation along `dim`, every other rank None"""
    group = _group_or_none()
    return input_ if group is None else group.gather(input_, dst, dim)


def broadcast_tensor_dict(tensor_dict: Optional[Dict[Any, Union[torch.Tensor, Any]]] = None, src: int = 0):
    group = _group_or_none()
    return tensor_dict if group is None else group.broadcast_tensor_dict(tensor_dict, src)
