"""`CustomAllreduce`: the communicator class an unmodified engine instantiates on the `_C_custom_ar` ops
(interface: reference vllm/distributed/device_communicators/custom_all_reduce.py:36-276 -- constructor arguments,
`disabled`, `capture()`, `custom_all_reduce()` returning None when it declines, `close()`).

How it works here.  Every rank owns three device tensors: `meta` (the kernel's flag block + the two-shot scratch),
`buffer` (an IPC-registered staging area for eager calls) and `rank_data` (pointer tables the kernel reads).  A
registration = "all-gather the IPC handle and offset of one of my tensors, hand the W handles to the kernel
library".  A call then runs in one of three modes:
  EAGER     copy the input into `buffer`, reduce out of the peers' buffers            (all_reduce_unreg)
  WARM-UP   inside `capture()` before the stream records: allocate the output only, so that the allocator sees the
            pattern the recorded step will have
  RECORDING inside `capture()` while the stream records: reduce straight out of the peers' INPUT tensors; their
            addresses are collected by the library and registered when the context closes (register_graph_buffers)
The decode harness of this repository does not use this class (it has the staging-buffer communicator of
../custom_all_reduce.py, fused with residual-add + RMSNorm); this is the drop-in form.

Platform notes.  Peer reachability comes from the HIP runtime (xGMI is a full mesh inside an MI355X node), not from
NVML.  IPC handles travel as latin-1 strings.  The flag block (`meta`) is an UNCACHED allocation of the library
(nmv_car_meta_alloc: hipExtMallocWithFlags(hipDeviceMallocUncached), as the staging communicator's), wrapped in a torch
tensor through `__cuda_array_interface__`: a peer's flag store and this rank's poll never sit in an L2 the other device
cannot see; the kernels' system-scope release / acquire order the payload.  It has still only RUN WITH ALL RANKS ON ONE
DEVICE, so the constructor keeps its self-test (eager one-shot and two-shot sizes against the rank-order sum computed on
the CPU, verdict agreed over the group) and disables itself on any mismatch: the caller then keeps the process group."""
import enum
from contextlib import contextmanager
from typing import Any, List, Optional, Sequence, Tuple, Union

import torch
import torch.distributed as dist
from torch.distributed import ProcessGroup

from ... import _custom_ops as ops
from ... import _lib


class _UncachedBlock:
    """owner of an uncached device allocation of libnmvllm_hip.so (nmv_car_meta_alloc), viewable as a uint8 tensor"""

    def __init__(self, nbytes: int, device: torch.device) -> None:
        import ctypes
        self.nbytes, self.device = nbytes, device
        ptr = ctypes.c_void_p()
        handle = ctypes.create_string_buffer(64)
        with torch.cuda.device(device):
            _lib.check(_lib.load().nmv_car_meta_alloc(nbytes, ctypes.byref(ptr), handle), "car_meta_alloc")
        self.ptr = int(ptr.value)
        self.ipc_handle = handle.raw.decode("latin-1")
        self.__cuda_array_interface__ = {"shape": (nbytes, ), "typestr": "|u1", "data": (self.ptr, False), "version": 2}

    def tensor(self) -> torch.Tensor:
        return torch.as_tensor(self, device=self.device)

    def free(self) -> None:
        if self.ptr:
            with torch.cuda.device(self.device):
                _lib.load().nmv_car_meta_free(self.ptr)
            self.ptr = 0


def _visible_physical_ids() -> List[int]:
    """physical device ids behind this process's ordinals 0 .. n-1: HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when set
    (the reference maps its peer check through CUDA_VISIBLE_DEVICES for the same reason, custom_all_reduce.py:23-33,
    _can_p2p), the identity otherwise; non-numeric entries (UUIDs) make the mapping unknown (empty list)"""
    import os
    n = torch.cuda.device_count()
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        raw_ids = os.environ.get(var)
        if raw_ids is not None and raw_ids.strip() != "":
            try:
                ids = [int(x) for x in raw_ids.split(",") if x.strip() != ""]
            except ValueError:
                return []
            return ids[:n]
    return list(range(n))


def _mesh_is_full(physical: Sequence[int], visible: Sequence[int]):
    """True / False: every pair of the ranks' PHYSICAL devices is (not) peer-reachable, asked of the runtime through this
    process's ordinals; None: some rank's device is not visible to this process (one device per rank), so the runtime
    cannot be asked here -- the caller then does not claim a full mesh.  Ranks that share a physical device (tests) are
    trivially reachable."""
    local = {p: i for i, p in enumerate(visible)}
    if any(p not in local for p in physical):
        return None
    return all(a == b or torch.cuda.can_device_access_peer(local[a], local[b]) for a in physical for b in physical)


class _Mode(enum.Enum):
    EAGER = 0
    CAPTURE_CONTEXT = 1     # warm-up or recording, told apart by the stream's state


class CustomAllreduce:

    _SUPPORTED_WORLD_SIZES = [2, 4, 6, 8]

    def __init__(self, group: ProcessGroup, device: Union[int, str, torch.device], max_size: int = 8192 * 1024) -> None:
        self.disabled = True
        self.disabled_reason = "single rank or unsupported world size"
        self._mode = _Mode.EAGER
        self._ptr = 0
        self.group = group
        assert dist.get_backend(group) != dist.Backend.NCCL, "CustomAllreduce should be attached to a non-NCCL group."
        self.rank, self.world_size = dist.get_rank(group=group), dist.get_world_size(group=group)
        if self.world_size not in self._SUPPORTED_WORLD_SIZES:
            return
        self.device = torch.device(f"cuda:{device}") if isinstance(device, int) else torch.device(device)
        self.max_size = max_size
        # local ordinals mean nothing across processes (with one visible device per rank every ordinal is 0): compare
        # physical ids, and when the peers' devices cannot be queried from here do not claim a full mesh
        visible = _visible_physical_ids()
        mine = visible[self.device.index] if self.device.index is not None and self.device.index < len(visible) else -1 - self.rank
        mesh = _mesh_is_full(self._everyones(mine), visible)
        self.full_nvlink = bool(mesh)
        if self.world_size > 2 and not self.full_nvlink:
            self.disabled_reason = ("peer reachability of the ranks' devices cannot be established from this process "
                                    "(device visibility differs per rank)" if mesh is None
                                    else "no peer access between every pair of devices")
            return
        def raw(nbytes: int, zeroed: bool = False) -> torch.Tensor:
            make = torch.zeros if zeroed else torch.empty
            return make(nbytes, dtype=torch.uint8, device=self.device)

        self._meta_block = _UncachedBlock(ops.meta_size() + max_size, self.device)   # flag block + two-shot scratch, zeroed
        self.meta = self._meta_block.tensor()
        self.buffer = raw(max_size)                                   # registered staging area of the eager path
        self.rank_data = raw(8 << 20)                                 # pointer tables
        torch.cuda.synchronize(self.device)              # the flag block is zero before a peer can map it
        handles, offsets = self._exchange((self._meta_block.ipc_handle, 0))
        self._ptr = ops.init_custom_ar(self.meta, self.rank_data, handles, offsets, self.rank, self.full_nvlink)
        self.disabled, self.disabled_reason = False, ""
        self.register_buffer(self.buffer)
        if not self._self_test():
            self.disabled = True

    # ---- IPC plumbing -------------------------------------------------------------------------------------
    def _everyones(self, mine: Any) -> List[Any]:
        out: List[Any] = [None] * self.world_size
        dist.all_gather_object(out, mine, group=self.group)
        return out

    @staticmethod
    def _ipc_of(t: torch.Tensor) -> Tuple[str, int]:
        shared = t.untyped_storage()._share_cuda_()
        return bytes(shared[1]).decode("latin-1"), int(shared[3])

    def _exchange(self, mine: Tuple[Any, Any]) -> Tuple[List[Any], List[Any]]:
        pairs = self._everyones(mine)
        return [p[0] for p in pairs], [p[1] for p in pairs]

    def register_buffer(self, inp: torch.Tensor):
        ops.register_buffer(self._ptr, inp, *self._exchange(self._ipc_of(inp)))

    def register_graph_buffers(self):
        handle, offsets = ops.get_graph_buffer_ipc_meta(self._ptr)
        ops.register_graph_buffers(self._ptr, *self._exchange((bytes(handle.numpy().tobytes()).decode("latin-1"),
                                                               list(offsets))))

    # ---- start-up self-test (see the module docstring) ------------------------------------------------------
    def _self_test(self) -> bool:
        ok = True
        try:
            for numel in (2048, 512 * 1024):     # 4 KB: one-shot; 1 MB: two-shot from 4 ranks on
                g = torch.Generator().manual_seed(1234 + numel)
                parts = [torch.randn(numel, generator=g).to(torch.bfloat16) for _ in range(self.world_size)]
                want = torch.stack([p.float() for p in parts]).sum(0).to(torch.bfloat16)    # rank order, one rounding
                got = self.all_reduce_unreg(parts[self.rank].to(self.device))
                torch.cuda.synchronize(self.device)
                ok = ok and torch.equal(got.cpu(), want)
        except Exception as e:   # a refused mapping, a timeout ...: never fatal, the process group carries on
            self.disabled_reason = f"self-test raised {type(e).__name__}: {e}"
            ok = False
        verdicts = self._everyones(bool(ok))
        if not all(verdicts):
            self.disabled_reason = self.disabled_reason or f"self-test mismatch on rank(s) {[r for r, v in enumerate(verdicts) if not v]}"
            return False
        return True

    # ---- the reference's public surface -----------------------------------------------------------------------
    @contextmanager
    def capture(self):
        self._mode = _Mode.CAPTURE_CONTEXT
        try:
            yield
        finally:
            self._mode = _Mode.EAGER
            if not self.disabled:
                self.register_graph_buffers()

    def should_custom_ar(self, inp: torch.Tensor):
        return ops.should_custom_ar(inp, self.max_size, self.world_size, self.full_nvlink)

    def _reduce(self, inp: torch.Tensor, out: Optional[torch.Tensor], via_staging: bool) -> torch.Tensor:
        result = out if out is not None else torch.empty_like(inp)
        if via_staging:
            ops.all_reduce_unreg(self._ptr, inp, self.buffer, result)
        else:
            ops.all_reduce_reg(self._ptr, inp, result)
        return result

    def all_reduce_reg(self, inp: torch.Tensor, out: Optional[torch.Tensor] = None):
        """`inp` is registered with the peers (register_buffer, or register_graph_buffers after a capture)"""
        return self._reduce(inp, out, via_staging=False)

    def all_reduce_unreg(self, inp: torch.Tensor, out: Optional[torch.Tensor] = None):
        """any tensor: it is copied into the registered staging buffer first"""
        return self._reduce(inp, out, via_staging=True)

    def custom_all_reduce(self, input: torch.Tensor) -> Optional[torch.Tensor]:
        """the reduced tensor, or None when this communicator does not take the message (disabled, too large, not a
        multiple of 16 bytes): the caller then uses the process group"""
        if self.disabled or not self.should_custom_ar(input):
            return None
        if self._mode is _Mode.EAGER:
            return self.all_reduce_unreg(input)
        if torch.cuda.is_current_stream_capturing():
            return self.all_reduce_reg(input)
        return torch.empty_like(input)           # warm-up inside capture(): the allocation pattern only

    def close(self):
        if self._ptr:
            ops.dispose(self._ptr)
            self._ptr = 0
        block = getattr(self, "_meta_block", None)
        if block is not None:      # after dispose: the library no longer touches the flag block
            self.meta = None
            block.free()
            self._meta_block = None

    def __del__(self):
        self.close()
