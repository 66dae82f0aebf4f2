from typing import Any, Dict, Optional

import torch


def set_weight_attrs(weight: torch.Tensor, weight_attrs: Optional[Dict[str, Any]]):
    """vllm/model_executor/utils.py:17-33: attach sharding attributes to a parameter."""
    if weight_attrs is None:
        return
    for key, value in weight_attrs.items():
        assert not hasattr(weight, key), f"Overwriting existing tensor attribute: {key}"
        setattr(weight, key, value)
