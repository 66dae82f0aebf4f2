"""Tensor-parallel group state: one process per GPU, torch.distributed over RCCL/xGMI.

Mirrors the part of vllm/distributed/parallel_state.py the hot path uses: GroupCoordinator
(:96-700) with all_reduce (:273-293), all_gather (:295-320), gather (:322-354),
broadcast_tensor_dict (:474-560 simplified) and the init helpers (:759-902).  Pipeline
parallelism, the shm broadcaster, pynccl graph capture and the CUDA-only custom all-reduce are
out of scope; on ROCm the reference itself always ends in
torch.distributed.all_reduce on the device group (:291), which is what this does.

The device group uses backend "nccl" (= RCCL on ROCm) when the tensors live on a GPU and "gloo"
on CPU, so the multi-rank logic is testable without GPUs (tests/test_distributed_cpu.py).
"""
from typing import Any, Dict, List, Optional, Union

import torch
import torch.distributed as dist


class GroupCoordinator:
    """A process group plus the collectives the model layers call."""

    def __init__(self, group_ranks: List[List[int]], local_rank: int, backend: str):
        self.rank = dist.get_rank()
        self.local_rank = local_rank
        self.device_group = None
        self.cpu_group = None
        for ranks in group_ranks:
            device_group = dist.new_group(ranks, backend=backend)
            # a gloo twin for CPU-side metadata, as the reference keeps (:144-148)
            cpu_group = dist.new_group(ranks, backend="gloo")
            if self.rank in ranks:
                self.ranks = ranks
                self.world_size = len(ranks)
                self.rank_in_group = ranks.index(self.rank)
                self.device_group = device_group
                self.cpu_group = cpu_group
        assert self.device_group is not None
        self.backend = backend
        if backend == "nccl" and torch.cuda.is_available():
            self.device = torch.device(f"cuda:{local_rank}")
        else:
            self.device = torch.device("cpu")
        # one-shot P2P all-reduce over HIP IPC for the decode-sized messages (custom_all_reduce.py);
        # None when disabled, unavailable or when its self-test did not pass on every rank
        self.custom_ar = None
        if self.world_size > 1 and torch.cuda.is_available():
            from .custom_all_reduce import maybe_create
            ar_dev = self.device if self.device.type == "cuda" else torch.device("cuda", torch.cuda.current_device())
            self.custom_ar = maybe_create(self.cpu_group, self.rank_in_group, self.world_size, ar_dev, backend)

    @property
    def first_rank(self):
        return self.ranks[0]

    @property
    def is_first_rank(self):
        return self.rank == self.first_rank

    def all_reduce(self, input_: torch.Tensor) -> torch.Tensor:
        """in-place sum all-reduce (parallel_state.py:273-293)"""
        if self.world_size == 1:
            return input_
        if self.custom_ar is not None and self.custom_ar.should_use(input_):
            return self.custom_ar.all_reduce(input_)
        dist.all_reduce(input_, group=self.device_group)
        return input_

    def all_gather(self, input_: torch.Tensor, dim: int = -1) -> torch.Tensor:
        world_size = self.world_size
        if world_size == 1:
            return input_
        assert -input_.dim() <= dim < input_.dim()
        if dim < 0:
            dim += input_.dim()
        input_size = input_.size()
        # flat [world*d0, ...] output: accepted by both RCCL and gloo
        inp = input_.contiguous() if input_.dim() > 0 else input_.reshape(1)
        flat = torch.empty((world_size * inp.shape[0], ) + tuple(inp.shape[1:]), dtype=inp.dtype,
                           device=inp.device)
        dist.all_gather_into_tensor(flat, inp, group=self.device_group)
        output_tensor = flat.reshape((world_size, ) + input_size).movedim(0, dim)
        return output_tensor.reshape(input_size[:dim] + (world_size * input_size[dim], ) +
                                     input_size[dim + 1:])

    def gather(self, input_: torch.Tensor, dst: int = 0, dim: int = -1) -> Optional[torch.Tensor]:
        """gather to group-rank `dst`; other ranks get None (parallel_state.py:322-354)"""
        world_size = self.world_size
        if world_size == 1:
            return input_
        if dim < 0:
            dim += input_.dim()
        gather_list = [torch.empty_like(input_) for _ in range(world_size)] \
            if self.rank_in_group == dst else None
        dist.gather(input_.contiguous(), gather_list, dst=self.ranks[dst], group=self.device_group)
        if self.rank_in_group == dst:
            return torch.cat(gather_list, dim=dim)
        return None

    def broadcast(self, input_: torch.Tensor, src: int = 0) -> torch.Tensor:
        if self.world_size == 1:
            return input_
        dist.broadcast(input_, src=self.ranks[src], group=self.device_group)
        return input_

    def broadcast_object(self, obj: Optional[Any] = None, src: int = 0) -> Any:
        if self.world_size == 1:
            return obj
        box = [obj]
        dist.broadcast_object_list(box, src=self.ranks[src], group=self.cpu_group)
        return box[0]

    def broadcast_tensor_dict(self, tensor_dict: Optional[Dict[str, Union[torch.Tensor, Any]]] = None,
                              src: int = 0) -> Optional[Dict[str, Union[torch.Tensor, Any]]]:
        """driver -> workers broadcast of the per-step inputs (worker_base.py:246-249): metadata
        over the gloo group, tensors over the device group."""
        if self.world_size == 1:
            return tensor_dict
        if self.rank_in_group == src:
            meta = []
            tensors = []
            for k, v in tensor_dict.items():
                if isinstance(v, torch.Tensor):
                    meta.append((k, ("tensor", v.dtype, tuple(v.shape), v.device.type)))
                    tensors.append(v)
                else:
                    meta.append((k, ("object", v)))
            self.broadcast_object(meta, src=src)
            for t in tensors:
                if t.numel():
                    g = self.cpu_group if t.device.type == "cpu" else self.device_group
                    dist.broadcast(t, src=self.ranks[src], group=g)
            return tensor_dict
        meta = self.broadcast_object(None, src=src)
        out: Dict[str, Any] = {}
        for k, m in meta:
            if m[0] == "tensor":
                _, dtype, shape, devtype = m
                device = self.device if devtype != "cpu" else torch.device("cpu")
                t = torch.empty(shape, dtype=dtype, device=device)
                if t.numel():
                    g = self.cpu_group if devtype == "cpu" else self.device_group
                    dist.broadcast(t, src=self.ranks[src], group=g)
                out[k] = t
            else:
                out[k] = m[1]
        return out

    def barrier(self):
        dist.barrier(group=self.cpu_group)

    def check_custom_ar_error(self, collective: bool = True) -> None:
        """raises CustomAllReduceError when a P2P all-reduce gave up waiting for a peer since the group
        was created (its output is NaN then); a no-op without the P2P communicator.  Synchronises."""
        if self.custom_ar is not None:
            self.custom_ar.check_error(collective)

    def destroy(self):
        pending = None
        if getattr(self, "custom_ar", None) is not None:
            try:
                # a timeout that nobody looked at must not pass silently; peers may already be gone
                self.custom_ar.check_error(collective=False)
            except Exception as e:   # finish the teardown first: the process groups must not leak
                pending = e
            finally:
                self.custom_ar.close()
                self.custom_ar = None
        if self.device_group is not None:
            dist.destroy_process_group(self.device_group)
            self.device_group = None
        if self.cpu_group is not None:
            dist.destroy_process_group(self.cpu_group)
            self.cpu_group = None
        if pending is not None:
            raise pending


_TP: Optional[GroupCoordinator] = None


def get_tp_group() -> GroupCoordinator:
    assert _TP is not None, "tensor model parallel group is not initialized"
    return _TP


def init_distributed_environment(world_size: int = -1, rank: int = -1,
                                 distributed_init_method: str = "env://", local_rank: int = -1,
                                 backend: str = "nccl") -> None:
    """parallel_state.py:759-799"""
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, init_method=distributed_init_method,
                                world_size=world_size, rank=rank)


def initialize_model_parallel(tensor_model_parallel_size: int = 1,
                              backend: Optional[str] = None, local_rank: int = 0) -> None:
    """parallel_state.py:832-902 restricted to TP (pipeline size 1)."""
    global _TP
    assert dist.is_initialized()
    world_size = dist.get_world_size()
    backend = backend or dist.get_backend()
    assert world_size % tensor_model_parallel_size == 0
    assert _TP is None, "tensor model parallel group is already initialized"
    group_ranks = [list(range(i * tensor_model_parallel_size, (i + 1) * tensor_model_parallel_size))
                   for i in range(world_size // tensor_model_parallel_size)]
    _TP = GroupCoordinator(group_ranks, local_rank, backend)


def ensure_model_parallel_initialized(tensor_model_parallel_size: int,
                                      backend: Optional[str] = None, local_rank: int = 0) -> None:
    if _TP is None:
        initialize_model_parallel(tensor_model_parallel_size, backend, local_rank)
        return
    assert _TP.world_size == tensor_model_parallel_size


def model_parallel_is_initialized() -> bool:
    return _TP is not None


def get_tensor_model_parallel_world_size() -> int:
    return _TP.world_size if _TP is not None else 1


def get_tensor_model_parallel_rank() -> int:
    return _TP.rank_in_group if _TP is not None else 0


def destroy_model_parallel() -> None:
    global _TP
    if _TP is not None:
        _TP.destroy()
    _TP = None
