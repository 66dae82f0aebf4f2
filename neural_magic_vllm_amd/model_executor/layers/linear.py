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


def adjust_marlin_shard(param, shard_size, shard_offset):
    marlin_tile_size = getattr(param, "marlin_tile_size", None)
    if marlin_tile_size is None:
        return shard_size, shard_offset
    return shard_size * marlin_tile_size, shard_offset * marlin_tile_size


def adjust_scalar_to_fused_array(param, loaded_weight, shard_id):
    """per-shard scalar scales of fused QKV / MLP modules (linear.py:49-66)"""
    qkv_idxs = {"q": 0, "k": 1, "v": 2}
    if isinstance(shard_id, str):
        shard_id = qkv_idxs[shard_id]
    elif not isinstance(shard_id, int):
        raise ValueError(f"Unknown Shard Id {shard_id}")
    if len(loaded_weight.shape) != 0:
        assert loaded_weight.shape[0] == 1
        loaded_weight = loaded_weight[0]
    return param[shard_id], loaded_weight


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

    def __init__(self, input_size: int, output_size: int, skip_bias_add: bool = False,
                 params_dtype: Optional[torch.dtype] = None,
                 quant_config: Optional[QuantizationConfig] = None):
        super().__init__()
        self.input_size = input_size
        self.output_size = output_size
        self.skip_bias_add = skip_bias_add
        if params_dtype is None:
            params_dtype = torch.get_default_dtype()
        self.params_dtype = params_dtype
        if quant_config is None:
            self.quant_method: Optional[QuantizeMethodBase] = UnquantizedLinearMethod()
        else:
            self.quant_method = quant_config.get_quant_method(self)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError


class ReplicatedLinear(LinearBase):

    def __init__(self, input_size, output_size, bias=True, skip_bias_add=False, params_dtype=None,
                 quant_config=None):
        super().__init__(input_size, output_size, skip_bias_add, params_dtype, quant_config)
        assert self.quant_method is not None
        self.quant_method.create_weights(self, self.input_size, [self.output_size],
                                         self.input_size, self.output_size, self.params_dtype)
        if bias:
            self.bias = Parameter(torch.empty(self.output_size, dtype=self.params_dtype))
            set_weight_attrs(self.bias, {"output_dim": 0})
        else:
            self.register_parameter("bias", None)

    def forward(self, x):
        bias = self.bias if not self.skip_bias_add else None
        output = self.quant_method.apply(self, x, bias)
        return output, (self.bias if self.skip_bias_add else None)


class ColumnParallelLinear(LinearBase):
    """Y = XA + b with A split along its output dimension: A = [A_1, ..., A_p]."""

    def __init__(self, input_size, output_size, bias=True, gather_output=False,
                 skip_bias_add=False, params_dtype=None, quant_config=None,
                 output_sizes: Optional[List[int]] = None):
        super().__init__(input_size, output_size, skip_bias_add, params_dtype, quant_config)
        self.gather_output = gather_output
        tp_size = get_tensor_model_parallel_world_size()
        assert self.quant_method is not None
        self.output_size_per_partition = divide(self.output_size, tp_size)
        self.output_partition_sizes = [self.output_size_per_partition]
        if hasattr(self, "output_sizes"):
            self.output_partition_sizes = [divide(s, tp_size) for s in self.output_sizes]
        if output_sizes is None:
            output_sizes = [output_size]
        self.quant_method.create_weights(self, self.input_size, self.output_partition_sizes,
                                         self.input_size, self.output_size, self.params_dtype,
                                         weight_loader=self.weight_loader)
        if bias:
            self.bias = Parameter(torch.empty(self.output_size_per_partition, dtype=params_dtype))
            set_weight_attrs(self.bias, {"output_dim": 0, "weight_loader": self.weight_loader})
        else:
            self.register_parameter("bias", None)

    def weight_loader(self, param: Parameter, loaded_weight: torch.Tensor):
        tp_rank = get_tensor_model_parallel_rank()
        output_dim = getattr(param, "output_dim", None)
        param_data = param.data
        if output_dim is not None:
            shard_size = param_data.shape[output_dim]
            loaded_weight = loaded_weight.narrow(output_dim, tp_rank * shard_size, shard_size)
        if len(loaded_weight.shape) == 0:
            loaded_weight = loaded_weight.reshape(1)
        assert param_data.shape == loaded_weight.shape
        param_data.copy_(loaded_weight)

    def forward_partial(self, input_):
        """Deferred split-K (not in the reference): the fp32 slabs [splits, T, N_partition] of X A for
        a consumer that sums them (ops.rotary_embedding_and_cache_partial), or None when this layer
        cannot defer (bias, gathered output, quantisation method / shape without the partial GEMM)."""
        qm = self.quant_method
        if self.bias is not None or self.gather_output or not hasattr(qm, "apply_partial") \
                or not isinstance(input_, torch.Tensor) or not input_.is_cuda or not qm.can_defer_reduce(self):
            return None
        rows = input_.numel() // input_.shape[-1]
        if ops.gptq_marlin_gemm_partial_splits(rows, self.output_size_per_partition, self.input_size) < 1:
            return None
        return qm.apply_partial(self, input_)

    def forward(self, input_):
        bias = self.bias if not self.skip_bias_add else None
        output_parallel = self.quant_method.apply(self, input_, bias)
        output = tensor_model_parallel_all_gather(output_parallel) if self.gather_output \
            else output_parallel
        return output, (self.bias if self.skip_bias_add else None)


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

    def weight_loader(self, param: Parameter, loaded_weight: torch.Tensor,
                      loaded_shard_id: Optional[int] = None):
        param_data = param.data
        output_dim = getattr(param, "output_dim", None)
        needs_scalar_to_array = getattr(param, "needs_scalar_to_array", None)
        if loaded_shard_id is None:
            # already-fused checkpoint tensor: load it shard by shard (linear.py:404-431)
            if output_dim is None:
                assert param_data.shape == loaded_weight.shape
                param_data.copy_(loaded_weight)
                return
            current = 0
            packed_dim = getattr(param, "packed_dim", None)
            for i, output_size in enumerate(self.output_sizes):
                shard_offset, shard_size = current, output_size
                current += output_size
                if packed_dim == output_dim:
                    shard_size //= param.pack_factor
                    shard_offset //= param.pack_factor
                    shard_size, shard_offset = adjust_marlin_shard(param, shard_size, shard_offset)
                self.weight_loader(param, loaded_weight.narrow(output_dim, shard_offset, shard_size), i)
            return
        assert loaded_shard_id < len(self.output_sizes)
        tp_rank = get_tensor_model_parallel_rank()
        tp_size = get_tensor_model_parallel_world_size()
        if output_dim is not None:
            shard_offset = sum(self.output_sizes[:loaded_shard_id]) // tp_size
            shard_size = self.output_sizes[loaded_shard_id] // tp_size
            packed_dim = getattr(param, "packed_dim", None)
            if packed_dim == output_dim:
                shard_size //= param.pack_factor
                shard_offset //= param.pack_factor
                shard_size, shard_offset = adjust_marlin_shard(param, shard_size, shard_offset)
            param_data = param_data.narrow(output_dim, shard_offset, shard_size)
            loaded_weight = loaded_weight.narrow(output_dim, tp_rank * shard_size, shard_size)
        elif needs_scalar_to_array is not None:
            param_data, loaded_weight = adjust_scalar_to_fused_array(param_data, loaded_weight,
                                                                     loaded_shard_id)
        assert param_data.shape == loaded_weight.shape
        param_data.copy_(loaded_weight)


class QKVParallelLinear(ColumnParallelLinear):
    """Fused Q/K/V projection; KV heads are replicated when tp_size > num_kv_heads."""

    def __init__(self, hidden_size, head_size, total_num_heads, total_num_kv_heads=None, bias=True,
                 skip_bias_add=False, params_dtype=None, quant_config=None):
        self.hidden_size = hidden_size
        self.head_size = head_size
        self.total_num_heads = total_num_heads
        if total_num_kv_heads is None:
            total_num_kv_heads = total_num_heads
        self.total_num_kv_heads = total_num_kv_heads
        tp_size = get_tensor_model_parallel_world_size()
        self.num_heads = divide(self.total_num_heads, tp_size)
        if tp_size >= self.total_num_kv_heads:
            self.num_kv_heads = 1
            self.num_kv_head_replicas = divide(tp_size, self.total_num_kv_heads)
        else:
            self.num_kv_heads = divide(self.total_num_kv_heads, tp_size)
            self.num_kv_head_replicas = 1
        input_size = self.hidden_size
        output_size = (self.num_heads + 2 * self.num_kv_heads) * tp_size * self.head_size
        self.output_sizes = [
            self.num_heads * self.head_size * tp_size,  # q_proj
            self.num_kv_heads * self.head_size * tp_size,  # k_proj
            self.num_kv_heads * self.head_size * tp_size,  # v_proj
        ]
        super().__init__(input_size=input_size, output_size=output_size, bias=bias,
                         gather_output=False, skip_bias_add=skip_bias_add,
                         params_dtype=params_dtype, quant_config=quant_config)

    def weight_loader(self, param: Parameter, loaded_weight: torch.Tensor,
                      loaded_shard_id: Optional[str] = None):
        param_data = param.data
        output_dim = getattr(param, "output_dim", None)
        needs_scalar_to_array = getattr(param, "needs_scalar_to_array", None)
        if loaded_shard_id is None:
            if output_dim is None:
                assert param_data.shape == loaded_weight.shape
                param_data.copy_(loaded_weight)
                return
            shard_offsets = [
                ("q", 0, self.total_num_heads * self.head_size),
                ("k", self.total_num_heads * self.head_size, self.total_num_kv_heads * self.head_size),
                ("v", (self.total_num_heads + self.total_num_kv_heads) * self.head_size,
                 self.total_num_kv_heads * self.head_size),
            ]
            packed_dim = getattr(param, "packed_dim", None)
            for shard_id, shard_offset, shard_size in shard_offsets:
                if packed_dim == output_dim:
                    shard_size //= param.pack_factor
                    shard_offset //= param.pack_factor
                    shard_size, shard_offset = adjust_marlin_shard(param, shard_size, shard_offset)
                self.weight_loader(param, loaded_weight.narrow(output_dim, shard_offset, shard_size),
                                   shard_id)
            return
        tp_rank = get_tensor_model_parallel_rank()
        assert loaded_shard_id in ["q", "k", "v"]
        if output_dim is not None:
            if loaded_shard_id == "q":
                shard_offset = 0
                shard_size = self.num_heads * self.head_size
            elif loaded_shard_id == "k":
                shard_offset = self.num_heads * self.head_size
                shard_size = self.num_kv_heads * self.head_size
            else:
                shard_offset = (self.num_heads + self.num_kv_heads) * self.head_size
                shard_size = self.num_kv_heads * self.head_size
            packed_dim = getattr(param, "packed_dim", None)
            if packed_dim == output_dim:
                shard_size //= param.pack_factor
                shard_offset //= param.pack_factor
                shard_size, shard_offset = adjust_marlin_shard(param, shard_size, shard_offset)
            param_data = param_data.narrow(output_dim, shard_offset, shard_size)
            shard_id = tp_rank if loaded_shard_id == "q" else tp_rank // self.num_kv_head_replicas
            loaded_weight = loaded_weight.narrow(output_dim, shard_id * shard_size, shard_size)
        elif needs_scalar_to_array is not None:
            param_data, loaded_weight = adjust_scalar_to_fused_array(param_data, loaded_weight,
                                                                     loaded_shard_id)
        assert param_data.shape == loaded_weight.shape
        param_data.copy_(loaded_weight)


class RowParallelLinear(LinearBase):
    """Y = XA + b with A split along its input dimension; partial results are all-reduced."""

    def __init__(self, input_size, output_size, bias=True, input_is_parallel=True,
                 skip_bias_add=False, params_dtype=None, reduce_results=True, quant_config=None):
        super().__init__(input_size, output_size, skip_bias_add, params_dtype, quant_config)
        self.input_is_parallel = input_is_parallel
        self.reduce_results = reduce_results
        self.defer_into_all_reduce = os.environ.get("NMV_FUSED_GLUE", "1") != "0"
        self.tp_size = get_tensor_model_parallel_world_size()
        self.input_size_per_partition = divide(input_size, self.tp_size)
        assert self.quant_method is not None
        self.quant_method.create_weights(self, self.input_size_per_partition, [self.output_size],
                                         self.input_size, self.output_size, self.params_dtype,
                                         weight_loader=self.weight_loader)
        if not reduce_results and (bias and not skip_bias_add):
            raise ValueError("When not reduce the results, adding bias to the results can lead "
                             "to incorrect results")
        if bias:
            self.bias = Parameter(torch.empty(self.output_size, dtype=params_dtype))
            set_weight_attrs(self.bias, {"output_dim": 0, "weight_loader": self.weight_loader})
        else:
            self.register_parameter("bias", None)

    def weight_loader(self, param: Parameter, loaded_weight: torch.Tensor):
        tp_rank = get_tensor_model_parallel_rank()
        input_dim = getattr(param, "input_dim", None)
        param_data = param.data
        if input_dim is not None:
            shard_size = param_data.shape[input_dim]
            loaded_weight = loaded_weight.narrow(input_dim, tp_rank * shard_size, shard_size)
        if len(loaded_weight.shape) == 0:
            loaded_weight = loaded_weight.reshape(1)
        assert param_data.shape == loaded_weight.shape
        param_data.copy_(loaded_weight)

    def forward_partial(self, input_):
        """Deferred split-K (not in the reference): the fp32 slabs [splits, T, N] of X A, for a
        consumer that sums them (ops.fused_add_rms_norm_partial), or None when this layer cannot
        defer -- a result that still has to be all-reduced or biased, or a quantisation method /
        shape without the partial GEMM."""
        if not self._can_defer(input_):
            return None
        if self.tp_size > 1:
            # under TP the slabs still have to be all-reduced: only worth leaving to the consumer
            # when the P2P communicator can take them (all_reduce_add_rms_norm)
            car = get_tp_group().custom_ar
            rows = input_.numel() // input_.shape[-1]
            if not self.reduce_results or car is None or not car.can_reduce(rows * self.output_size):
                return None
        return self.quant_method.apply_partial(self, input_)

    def _can_defer(self, input_) -> bool:
        qm = self.quant_method
        if self.bias is not None or not self.input_is_parallel or not hasattr(qm, "apply_partial") \
                or not qm.can_defer_reduce(self) or not input_.is_cuda \
                or self.output_size % 8 != 0 or self.output_size > 8192:
            return False
        rows = input_.numel() // input_.shape[-1]
        return ops.gptq_marlin_gemm_partial_splits(rows, self.output_size, self.input_size_per_partition) >= 1

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
        output_parallel = self.quant_method.apply(self, input_parallel)
        if self.reduce_results and self.tp_size > 1:
            output_ = tensor_model_parallel_all_reduce(output_parallel)  # RCCL over xGMI
        else:
            output_ = output_parallel
        if not self.skip_bias_add:
            output = output_ + self.bias if self.bias is not None else output_
            output_bias = None
        else:
            output = output_
            output_bias = self.bias
        return output, output_bias
