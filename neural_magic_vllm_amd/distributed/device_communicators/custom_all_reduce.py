"""The reference's `CustomAllreduce` communicator (vllm/distributed/device_communicators/custom_all_reduce.py
:36-276) on the `_C_custom_ar` ops of this package: registered IPC buffers, an eager path that copies into
the pre-registered buffer (all_reduce_unreg), a graph path that records the graph-private inputs and
registers them after the capture (capture() -> register_graph_buffers), one-shot / two-shot by size.

This is the drop-in form for an unmodified engine.  The decode harness of this repo uses the staging-buffer
communicator in ../custom_all_reduce.py instead (fused with residual-add + RMSNorm, no registration step).

Differences from the reference, all forced by the platform: the topology test asks the HIP runtime for peer
access between every pair of devices (xGMI is a full mesh on an MI355X node) where the reference asks NVML
for NVLink; IPC handles are carried as latin-1 strings."""
from contextlib import contextmanager
from typing import Any, List, Optional, Union

import torch
import torch.distributed as dist
from torch.distributed import ProcessGroup

from ... import _custom_ops as ops


def _full_peer_access(device_ids: List[int]) -> bool:
    """every pair of the group's devices can map each other's memory (is_full_nvlink, :22-33)"""
    for i in device_ids:
        for j in device_ids:
            if i != j and not torch.cuda.can_device_access_peer(i, j):
                return False
    return True


class CustomAllreduce:

    _SUPPORTED_WORLD_SIZES = [2, 4, 6, 8]

    def __init__(self, group: ProcessGroup, device: Union[int, str, torch.device], max_size: int = 8192 * 1024) -> None:
        self._IS_CAPTURING = False
        self.disabled = True
        self.group = group
        assert dist.get_backend(group) != dist.Backend.NCCL, "CustomAllreduce should be attached to a non-NCCL group."
        rank = dist.get_rank(group=self.group)
        world_size = dist.get_world_size(group=self.group)
        if world_size == 1 or world_size not in self._SUPPORTED_WORLD_SIZES:
            return
        if isinstance(device, int):
            device = torch.device(f"cuda:{device}")
        elif isinstance(device, str):
            device = torch.device(device)
        self.device = device
        ids: List[Any] = [None] * world_size
        dist.all_gather_object(ids, device.index, group=self.group)
        full_link = _full_peer_access([i for i in ids if i is not None])
        if world_size > 2 and not full_link:
            return
        self.disabled = False
        # synchronisation words + scratch of the two-shot form; the pre-registered eager buffer; pointer tables
        self.meta = torch.zeros(ops.meta_size() + max_size, dtype=torch.uint8, device=self.device)
        self.buffer = torch.empty(max_size, dtype=torch.uint8, device=self.device)
        self.rank_data = torch.empty(8 * 1024 * 1024, dtype=torch.uint8, device=self.device)
        self.max_size = max_size
        self.rank = rank
        self.world_size = world_size
        self.full_nvlink = full_link
        torch.cuda.synchronize(self.device)     # meta is zero before any peer maps it
        handles, offsets = self._get_ipc_meta(self.meta)
        self._ptr = ops.init_custom_ar(self.meta, self.rank_data, handles, offsets, rank, self.full_nvlink)
        self.register_buffer(self.buffer)

    @contextmanager
    def capture(self):
        """graph capture: inputs seen inside are recorded and registered when the context ends (:183-197)"""
        try:
            self._IS_CAPTURING = True
            yield
        finally:
            self._IS_CAPTURING = False
            if not self.disabled:
                self.register_graph_buffers()

    def _get_ipc_meta(self, inp: torch.Tensor):
        data = inp.untyped_storage()._share_cuda_()
        return self._gather_ipc_meta((bytes(data[1]).decode("latin-1"), int(data[3])))

    def _gather_ipc_meta(self, shard_data):
        all_data: List[Any] = [None] * self.world_size
        dist.all_gather_object(all_data, shard_data, group=self.group)
        return [d[0] for d in all_data], [d[1] for d in all_data]

    def register_buffer(self, inp: torch.Tensor):
        handles, offsets = self._get_ipc_meta(inp)
        ops.register_buffer(self._ptr, inp, handles, offsets)

    def register_graph_buffers(self):
        handle, offset = ops.get_graph_buffer_ipc_meta(self._ptr)
        handles, offsets = self._gather_ipc_meta((bytes(handle.numpy().tobytes()).decode("latin-1"), list(offset)))
        ops.register_graph_buffers(self._ptr, handles, offsets)

    def should_custom_ar(self, inp: torch.Tensor):
        return ops.should_custom_ar(inp, self.max_size, self.world_size, self.full_nvlink)

    def all_reduce_reg(self, inp: torch.Tensor, out: Optional[torch.Tensor] = None):
        """inp is IPC-registered (register_buffer, or register_graph_buffers after a capture)"""
        if out is None:
            out = torch.empty_like(inp)
        ops.all_reduce_reg(self._ptr, inp, out)
        return out

    def all_reduce_unreg(self, inp: torch.Tensor, out: Optional[torch.Tensor] = None):
        if out is None:
            out = torch.empty_like(inp)
        ops.all_reduce_unreg(self._ptr, inp, self.buffer, out)
        return out

    def custom_all_reduce(self, input: torch.Tensor) -> Optional[torch.Tensor]:
        """None when the message is not taken (disabled, too large, not 16-byte sized): the caller falls back
        to the process group (:249-270)"""
        if self.disabled:
            return None
        if self._IS_CAPTURING:
            if torch.cuda.is_current_stream_capturing():
                if self.should_custom_ar(input):
                    return self.all_reduce_reg(input)
            elif self.should_custom_ar(input):
                return torch.empty_like(input)    # warm-up: mimic the allocation pattern
        elif self.should_custom_ar(input):
            return self.all_reduce_unreg(input)
        return None

    def close(self):
        if not self.disabled and getattr(self, "_ptr", 0):
            ops.dispose(self._ptr)
            self._ptr = 0

    def __del__(self):
        self.close()
