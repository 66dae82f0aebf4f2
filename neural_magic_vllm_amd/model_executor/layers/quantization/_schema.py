"""Declarative parameter schemas for the linear methods of this package.

A quantisation method states, as data, (1) the conditions a shard must satisfy and (2) the parameters it
owns -- name, shape, dtype, fill, sharding attributes -- as functions of the shard's geometry.  `build()`
checks and materialises them on a layer.  The tables are pinned against the reference's own
`create_weights` by tests/golden/linear_method_params.json (tests/test_linear_methods_cpu.py): names, shapes,
dtypes and attributes must match what an unmodified loader expects (vllm/model_executor/layers/linear.py
weight_loader contracts)."""
from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple, Union

import torch
from torch.nn.parameter import Parameter

from ...utils import set_weight_attrs


@dataclass(frozen=True)
class Geometry:
    """what a linear method is told about the shard it creates weights for"""
    k: int                    # input features on this rank
    parts: Tuple[int, ...]    # output features of each logical matrix on this rank
    dtype: torch.dtype        # model dtype
    k_total: Optional[int] = None   # input features across all ranks (None: same as k)

    @property
    def n(self) -> int:
        return sum(self.parts)

    @property
    def k_all(self) -> int:
        return self.k if self.k_total is None else self.k_total

    @property
    def row_sharded(self) -> bool:
        """this rank holds a slice of the input dimension (a row-parallel layer under TP > 1)"""
        return self.k_all != self.k


@dataclass(frozen=True)
class Require:
    """a condition on the geometry and the ValueError text when it does not hold"""
    holds: Callable[[Geometry], bool]
    message: Callable[[Geometry], str]


@dataclass(frozen=True)
class Slot:
    """one parameter of the layer"""
    name: str
    shape: Callable[[Geometry], Sequence[int]]
    dtype: Union[torch.dtype, Callable[[Geometry], torch.dtype]]
    attrs: Callable[[Geometry], Dict[str, Any]] = lambda g: {}
    fill: Optional[Union[str, float]] = None      # None: uninitialised, "zeros", or a value
    init: Optional[Callable[[Geometry], torch.Tensor]] = None   # the initial contents, when they are not a constant
    device: Optional[str] = None                  # "meta": declared for the loader's sake, never materialised


def build(layer: torch.nn.Module, geometry: Geometry, requires: List[Require], slots: List[Slot],
          loader_attrs: Optional[Dict[str, Any]] = None) -> None:
    for r in requires:
        if not r.holds(geometry):
            raise ValueError(r.message(geometry))
    for s in slots:
        dt = s.dtype(geometry) if callable(s.dtype) else s.dtype
        shape = tuple(s.shape(geometry))
        if s.init is not None:
            data = s.init(geometry).to(dt)
            assert tuple(data.shape) == shape, (s.name, tuple(data.shape), shape)
        else:
            data = (torch.zeros(shape, dtype=dt, device=s.device) if s.fill == "zeros"
                    else torch.empty(shape, dtype=dt, device=s.device))
            if s.fill is not None and s.fill != "zeros":
                data[...] = s.fill
        p = Parameter(data, requires_grad=False)
        layer.register_parameter(s.name, p)
        set_weight_attrs(p, s.attrs(geometry))
        set_weight_attrs(p, loader_attrs)
