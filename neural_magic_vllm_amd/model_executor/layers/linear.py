"""Tensor-parallel linear layers and the LinearMethodBase plugin point.

Mirror of vllm/model_executor/layers/linear.py: LinearMethodBase (:69-100),
UnquantizedLinearMethod (:103-136), LinearBase (:139-175), ReplicatedLinear (:178-226),
ColumnParallelLinear (:229-357), MergedColumnParallelLinear (:360-491), QKVParallelLinear
(:494-699), RowParallelLinear (:702-811).  Same constructor arguments, same parameter
attributes (input_dim / output_dim / packed_dim / pack_factor) and the same weight_loader
sharding rules, so quantisation methods written for the reference plug in unchanged.  The
row-parallel all-reduce goes through RCCL (distributed/communication_op.py).
"""
import os
from typing import List, Optional, Tuple

import torch
import torch.nn.functional as F
from torch.nn.parameter import Parameter

from ... import _custom_ops as ops
from ...distributed import (get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size,
                            get_tp_group, tensor_model_parallel_all_gather,
                            tensor_model_parallel_all_reduce)
from ..utils import set_weight_attrs
from .quantization.base_config import (LinearMethodBase, QuantizationConfig,  # noqa: F401
                                       QuantizeMethodBase)


def divide(numerator: int, denominator: int) -> int:
    assert numerator % denominator == 0, f"{numerator} is not divisible by {denominator}"
    return numerator // denominator


# ---- checkpoint tensors -> (fused, sharded, maybe packed) parameters ------------------------------------
# Every parallel linear describes its parameter's output dimension as a list of LOGICAL shards (one for a
# plain layer, gate | up for the merged MLP projection, q | k | v for attention) and loads through ONE routine,
# `_place`.  A parameter states its own geometry through attributes set by the quantisation method
# (reference contract, linear.py:29-66, 268-287, 404-491, 567-699, 765-782): output_dim / input_dim, packed_dim +
# pack_factor (several logical columns per stored element), marlin_tile_size (a Marlin tensor stores tile_size
# logical columns per unit of its packed dimension the other way round), needs_scalar_to_array (one scalar per
# logical shard of a fused module).

def _method_can_defer(qm, layer, rows: int, n: int, k: int) -> bool:
    """one plan query for the admission check and the launch: the method's own can_defer(layer, rows) (it knows the tensor
    and the group count apply_partial will use); methods without the hook are asked through the generic splits query with
    the layer's real group count"""
    hook = getattr(qm, "can_defer", None)
    if hook is not None:
        return bool(hook(layer, rows))
    scales = getattr(layer, "scales", None)
    groups = int(scales.shape[0]) if scales is not None and scales.dim() == 2 else None
    return ops.gptq_marlin_gemm_partial_splits(rows, n, k, groups) >= 1

class _Shard:
    """one logical sub-matrix of a fused output dimension: where it sits in this rank's parameter (`local`), in
    the un-sharded checkpoint tensor (`whole`), and which slice of a per-matrix checkpoint tensor this rank takes
    (`source`: the rank, or the KV-head group when KV heads are replicated)"""
    __slots__ = ("key", "index", "local_offset", "local_size", "whole_offset", "whole_size", "source")

    def __init__(self, key, index, local_offset, local_size, whole_offset, whole_size, source):
        self.key, self.index = key, index
        self.local_offset, self.local_size = local_offset, local_size
        self.whole_offset, self.whole_size = whole_offset, whole_size
        self.source = source


def _stored_units(param, columns: int) -> int:
    """logical output columns -> units of the parameter's output dimension"""
    if getattr(param, "packed_dim", None) == getattr(param, "output_dim", None):
        columns //= param.pack_factor
        tile = getattr(param, "marlin_tile_size", None)
        if tile is not None:
            columns *= tile
    return columns


def _copy_checked(dst: torch.Tensor, src: torch.Tensor) -> None:
    if src.numel() == 1 and dst.numel() == 1:    # a scalar saved as [] or [1] into a [] or [1] slot
        src = src.reshape(dst.shape)
    assert dst.shape == src.shape, f"checkpoint tensor {tuple(src.shape)} does not fit parameter {tuple(dst.shape)}"
    dst.copy_(src)


def _place(param: Parameter, loaded: torch.Tensor, shard: "_Shard") -> None:
    """one per-matrix checkpoint tensor (q_proj.weight, gate_proj.qweight, a scalar weight_scale ...) into its
    place in the fused parameter"""
    axis = getattr(param, "output_dim", None)
    if axis is not None:
        size = _stored_units(param, shard.local_size)
        dst = param.data.narrow(axis, _stored_units(param, shard.local_offset), size)
        _copy_checked(dst, loaded.narrow(axis, shard.source * size, size))
    elif getattr(param, "needs_scalar_to_array", None) is not None:
        if loaded.dim() != 0:            # [1] -> scalar
            assert loaded.shape[0] == 1
            loaded = loaded[0]
        _copy_checked(param.data[shard.index], loaded)
    else:
        _copy_checked(param.data, loaded)


def _place_fused(layer, param: Parameter, loaded: torch.Tensor, shards) -> None:
    """an already-fused checkpoint tensor (qkv_proj / gate_up_proj saved as one): cut it at the shards' positions
    in the WHOLE output dimension and place the pieces one by one"""
    axis = getattr(param, "output_dim", None)
    if axis is None:
        _copy_checked(param.data, loaded)
        return
    for sh in shards:
        piece = loaded.narrow(axis, _stored_units(param, sh.whole_offset), _stored_units(param, sh.whole_size))
        layer.weight_loader(param, piece, sh.key)


class UnquantizedLinearMethod(LinearMethodBase):
    """Plain library GEMM (hipBLASLt through F.linear) -- used for lm_head / bf16 configs."""

    def __init__(self, separate_bias_add: bool = False):
        self.separate_bias_add = separate_bias_add

    def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size,
                       output_size, params_dtype, **extra_weight_attrs):
        weight = Parameter(torch.empty(sum(output_partition_sizes), input_size_per_partition,
                                       dtype=params_dtype), requires_grad=False)
        set_weight_attrs(weight, {"input_dim": 1, "output_dim": 0})
        layer.register_parameter("weight", weight)
        set_weight_attrs(weight, extra_weight_attrs)

    def apply(self, layer, x, bias=None):
        if self.separate_bias_add:
            if bias is not None:
                return F.linear(x, layer.weight) + bias
            return F.linear(x, layer.weight)
        return F.linear(x, layer.weight, bias)


class LinearBase(torch.nn.Module):
    """common state of every linear layer: sizes, dtype, and the quantisation method that owns the weights
    (quant_config.get_quant_method(self); the plain library GEMM without a config)"""

    def __init__(self, input_size: int, output_size: int, skip_bias_add: bool = False,
                 params_dtype: Optional[torch.dtype] = None,
                 quant_config: Optional[QuantizationConfig] = None):
        super().__init__()
        self.input_size, self.output_size = input_size, output_size
        self.skip_bias_add = skip_bias_add
        self.params_dtype = params_dtype if params_dtype is not None else torch.get_default_dtype()
        self.quant_method: Optional[QuantizeMethodBase] = (
            UnquantizedLinearMethod() if quant_config is None else quant_config.get_quant_method(self))

    def _own_bias(self, wanted: bool, width: int, loader=None) -> None:
        """`bias` [width] in the model dtype (sharded like the output dimension), or a registered None"""
        if not wanted:
            self.register_parameter("bias", None)
            return
        self.bias = Parameter(torch.empty(width, dtype=self.params_dtype))
        set_weight_attrs(self.bias, {"output_dim": 0} if loader is None else {"output_dim": 0, "weight_loader": loader})

    def _split_bias(self):
        """(bias for the GEMM epilogue, bias handed back to the caller) under skip_bias_add"""
        return (None, self.bias) if self.skip_bias_add else (self.bias, None)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError


class ReplicatedLinear(LinearBase):

    def __init__(self, input_size, output_size, bias=True, skip_bias_add=False, params_dtype=None,
                 quant_config=None):
        super().__init__(input_size, output_size, skip_bias_add, params_dtype, quant_config)
        assert self.quant_method is not None
        self.quant_method.create_weights(self, input_size, [output_size], input_size, output_size, self.params_dtype)
        self._own_bias(bias, output_size)

    def forward(self, x):
        fused_bias, returned_bias = self._split_bias()
        return self.quant_method.apply(self, x, fused_bias), returned_bias


class ColumnParallelLinear(LinearBase):
    """Y = XA + b with A split along its output dimension: A = [A_1, ..., A_p]."""

    def __init__(self, input_size, output_size, bias=True, gather_output=False,
                 skip_bias_add=False, params_dtype=None, quant_config=None,
                 output_sizes: Optional[List[int]] = None):
        super().__init__(input_size, output_size, skip_bias_add, params_dtype, quant_config)
        assert self.quant_method is not None
        self.gather_output = gather_output
        tp = get_tensor_model_parallel_world_size()
        # a fused subclass has set self.output_sizes (the logical matrices) before calling up
        logical = getattr(self, "output_sizes", None) or [self.output_size]
        self.output_partition_sizes = [divide(n, tp) for n in logical]
        self.output_size_per_partition = divide(self.output_size, tp)
        self.quant_method.create_weights(self, self.input_size, self.output_partition_sizes, self.input_size,
                                         self.output_size, self.params_dtype, weight_loader=self.weight_loader)
        self._own_bias(bias, self.output_size_per_partition, self.weight_loader)

    def weight_loader(self, param: Parameter, loaded_weight: torch.Tensor):
        width = self.output_size_per_partition
        _place(param, loaded_weight, _Shard(None, 0, 0, width, 0, self.output_size, get_tensor_model_parallel_rank()))

    def forward_partial(self, input_, allow16: bool = False):
        """Deferred split-K (not in the reference): the fp32 slabs [splits, T, N_partition] of X A for
        a consumer that sums them (ops.rotary_embedding_and_cache_partial), or None when this layer
        cannot defer (bias, gathered output, quantisation method / shape without the partial GEMM).
        allow16: the consumer also reads slabs in the model dtype (prompt steps: LinearMethod.apply_partial)."""
        qm = self.quant_method
        if self.bias is not None or self.gather_output or not hasattr(qm, "apply_partial") \
                or not isinstance(input_, torch.Tensor) or not input_.is_cuda or not qm.can_defer_reduce(self):
            return None
        rows = input_.numel() // input_.shape[-1]
        if not _method_can_defer(qm, self, rows, self.output_size_per_partition, self.input_size):
            return None
        return qm.apply_partial(self, input_, allow16) if allow16 else qm.apply_partial(self, input_)

    def forward(self, input_):
        fused_bias, returned_bias = self._split_bias()
        y = self.quant_method.apply(self, input_, fused_bias)
        return (tensor_model_parallel_all_gather(y) if self.gather_output else y), returned_bias


class MergedColumnParallelLinear(ColumnParallelLinear):
    """Several column-parallel matrices packed along the output dimension (gate_up_proj)."""

    def __init__(self, input_size, output_sizes: List[int], bias=True, gather_output=False,
                 skip_bias_add=False, params_dtype=None, quant_config=None):
        self.output_sizes = output_sizes
        tp_size = get_tensor_model_parallel_world_size()
        assert all(s % tp_size == 0 for s in output_sizes)
        super().__init__(input_size=input_size, output_size=sum(output_sizes), bias=bias,
                         gather_output=gather_output, skip_bias_add=skip_bias_add,
                         params_dtype=params_dtype, quant_config=quant_config)

    def _shards(self):
        tp, rank = get_tensor_model_parallel_world_size(), get_tensor_model_parallel_rank()
        out, local, whole = [], 0, 0
        for i, n in enumerate(self.output_sizes):
            out.append(_Shard(i, i, local, n // tp, whole, n, rank))
            local, whole = local + n // tp, whole + n
        return out

    def weight_loader(self, param: Parameter, loaded_weight: torch.Tensor,
                      loaded_shard_id: Optional[int] = None):
        shards = self._shards()
        if loaded_shard_id is None:
            _place_fused(self, param, loaded_weight, shards)
            return
        assert loaded_shard_id < len(shards)
        _place(param, loaded_weight, shards[loaded_shard_id])


class QKVParallelLinear(ColumnParallelLinear):
    """Fused Q/K/V projection; KV heads are replicated when tp_size > num_kv_heads."""

    def __init__(self, hidden_size, head_size, total_num_heads, total_num_kv_heads=None, bias=True,
                 skip_bias_add=False, params_dtype=None, quant_config=None):
        tp = get_tensor_model_parallel_world_size()
        self.hidden_size, self.head_size = hidden_size, head_size
        self.total_num_heads = total_num_heads
        self.total_num_kv_heads = total_num_heads if total_num_kv_heads is None else total_num_kv_heads
        self.num_heads = divide(total_num_heads, tp)
        # KV heads: split over the ranks while there are enough of them, else one per rank, each shared by
        # tp / total_num_kv_heads consecutive ranks
        replicated = tp >= self.total_num_kv_heads
        self.num_kv_heads = 1 if replicated else divide(self.total_num_kv_heads, tp)
        self.num_kv_head_replicas = divide(tp, self.total_num_kv_heads) if replicated else 1
        q_width = self.num_heads * head_size * tp
        kv_width = self.num_kv_heads * head_size * tp      # counts a replicated head once per rank
        self.output_sizes = [q_width, kv_width, kv_width]
        super().__init__(input_size=hidden_size, output_size=q_width + 2 * kv_width, bias=bias, gather_output=False,
                         skip_bias_add=skip_bias_add, params_dtype=params_dtype, quant_config=quant_config)

    def _shards(self):
        """q | k | v: a rank owns num_heads query heads and num_kv_heads KV heads; with more ranks than KV heads a
        KV head is replicated over num_kv_head_replicas consecutive ranks, which all read the same slice"""
        rank = get_tensor_model_parallel_rank()
        d = self.head_size
        q_local, kv_local = self.num_heads * d, self.num_kv_heads * d
        q_whole, kv_whole = self.total_num_heads * d, self.total_num_kv_heads * d
        kv_source = rank // self.num_kv_head_replicas
        return {"q": _Shard("q", 0, 0, q_local, 0, q_whole, rank),
                "k": _Shard("k", 1, q_local, kv_local, q_whole, kv_whole, kv_source),
                "v": _Shard("v", 2, q_local + kv_local, kv_local, q_whole + kv_whole, kv_whole, kv_source)}

    def weight_loader(self, param: Parameter, loaded_weight: torch.Tensor,
                      loaded_shard_id: Optional[str] = None):
        shards = self._shards()
        if loaded_shard_id is None:
            _place_fused(self, param, loaded_weight, shards.values())
            return
        assert loaded_shard_id in shards, f"Unknown Shard Id {loaded_shard_id}"
        _place(param, loaded_weight, shards[loaded_shard_id])


class RowParallelLinear(LinearBase):
    """Y = XA + b with A split along its input dimension; partial results are all-reduced."""

    def __init__(self, input_size, output_size, bias=True, input_is_parallel=True,
                 skip_bias_add=False, params_dtype=None, reduce_results=True, quant_config=None):
        super().__init__(input_size, output_size, skip_bias_add, params_dtype, quant_config)
        assert self.quant_method is not None
        if bias and not skip_bias_add and not reduce_results:
            raise ValueError("When not reduce the results, adding bias to the results can lead "
                             "to incorrect results")
        self.input_is_parallel, self.reduce_results = input_is_parallel, reduce_results
        self.defer_into_all_reduce = os.environ.get("NMV_FUSED_GLUE", "1") != "0"
        self.tp_size = get_tensor_model_parallel_world_size()
        self.input_size_per_partition = divide(input_size, self.tp_size)
        self.quant_method.create_weights(self, self.input_size_per_partition, [output_size], input_size, output_size,
                                         self.params_dtype, weight_loader=self.weight_loader)
        self._own_bias(bias, output_size, self.weight_loader)

    def weight_loader(self, param: Parameter, loaded_weight: torch.Tensor):
        axis = getattr(param, "input_dim", None)
        if axis is not None:       # this rank's slice of the input dimension (packed parameters are cut in their own units)
            rows = param.data.shape[axis]
            loaded_weight = loaded_weight.narrow(axis, get_tensor_model_parallel_rank() * rows, rows)
        _copy_checked(param.data, loaded_weight)

    def forward_partial(self, input_, allow16: bool = False):
        """Deferred split-K (not in the reference): the fp32 slabs [splits, T, N] of X A, for a
        consumer that sums them (ops.fused_add_rms_norm_partial), or None when this layer cannot
        defer -- a result that still has to be all-reduced or biased, or a quantisation method /
        shape without the partial GEMM.  allow16: the consumer also reads slabs in the model dtype (the norm launch does;
        the all-reduce of a tensor-parallel group does not)."""
        if not self._can_defer(input_):
            return None
        if self.tp_size > 1:
            # under TP the slabs still have to be all-reduced: only worth leaving to the consumer
            # when the P2P communicator can take them (all_reduce_add_rms_norm)
            car = get_tp_group().custom_ar
            rows = input_.numel() // input_.shape[-1]
            if not self.reduce_results or car is None or not car.can_reduce(rows * self.output_size):
                return None
        if allow16 and self.tp_size == 1:
            return self.quant_method.apply_partial(self, input_, True)
        return self.quant_method.apply_partial(self, input_)

    def _can_defer(self, input_) -> bool:
        qm = self.quant_method
        if self.bias is not None or not self.input_is_parallel or not hasattr(qm, "apply_partial") \
                or not qm.can_defer_reduce(self) or not input_.is_cuda \
                or self.output_size % 8 != 0 or self.output_size > 8192:
            return False
        rows = input_.numel() // input_.shape[-1]
        return _method_can_defer(qm, self, rows, self.output_size, self.input_size_per_partition)

    def forward(self, input_):
        if self.input_is_parallel:
            input_parallel = input_
        else:
            tp_rank = get_tensor_model_parallel_rank()
            input_parallel = torch.chunk(input_, self.tp_size, dim=-1)[tp_rank].contiguous()
        assert self.quant_method is not None
        if self.reduce_results and self.tp_size > 1 and self.defer_into_all_reduce:
            # the P2P all-reduce sums this rank's fp32 split-K slabs itself (deferred reduction under TP)
            car = get_tp_group().custom_ar
            rows = input_parallel.numel() // input_parallel.shape[-1]
            if car is not None and car.can_reduce(rows * self.output_size) and self._can_defer(input_parallel):
                slab = self.quant_method.apply_partial(self, input_parallel)
                output_ = car.all_reduce_partial(slab, input_parallel.dtype).reshape(
                    input_parallel.shape[:-1] + (self.output_size, ))
                return output_, None
        y = self.quant_method.apply(self, input_parallel)
        if self.reduce_results and self.tp_size > 1:
            y = tensor_model_parallel_all_reduce(y)   # RCCL over xGMI (or the P2P kernels)
        # the bias is added once, after the reduction (every rank holds the whole bias)
        added, returned_bias = self._split_bias()
        return (y if added is None else y + added), returned_bias
