"""One-shot / two-shot P2P all-reduce over HIP IPC (csrc/custom_all_reduce.hip) -- the ROCm analogue of
the reference's vllm/distributed/device_communicators/custom_all_reduce.py (CUDA only there; its
kernels are compiled out on ROCm, csrc/torch_bindings.cpp:261).

Every rank allocates one staging block on its GPU, the IPC handles are exchanged over the gloo
twin of the tensor-parallel group, and `all_reduce` launches one kernel per call (capturable into
a hipGraph: the call counters live in device memory).  Before the communicator is used it runs a
self-test against a CPU reference on the actual devices; any mismatch, a bounded-spin timeout or a
failed IPC mapping on ANY rank disables it on ALL ranks and RCCL is used instead -- the custom
path can make the decode step faster, never wrong.  After start-up a flag wait that runs out (a lost
or lagging peer) makes the call write NaN and sets an error word on the device: `check_error` reads
it at the points that synchronise anyway (after graph capture, after a timed loop, on destroy) and
raises `CustomAllReduceError` on every rank."""
import ctypes
import os
import socket
from typing import List, Optional

import torch
import torch.distributed as dist

from .. import _lib
from .._torch_bindings import check, dtype_code, ptr, stream_of

MAX_BYTES = 8 << 20  # the reference's max_size (custom_all_reduce.py:43)


class CustomAllReduceError(RuntimeError):
    """a flag wait of the P2P all-reduce timed out on some rank: results since then are NaN"""


class CustomAllReduce:

    def __init__(self, cpu_group, rank_in_group: int, world_size: int, device: torch.device,
                 max_bytes: int = MAX_BYTES) -> None:
        self.cpu_group, self.rank, self.world, self.device = cpu_group, rank_in_group, world_size, device
        self.max_bytes = max_bytes
        self.state = None
        self.disabled_reason: Optional[str] = None
        self.enabled = False
        lib = _lib.load()
        ok, handle = True, b""
        try:
            hb = lib.nmv_ar_handle_bytes()
            buf = ctypes.create_string_buffer(hb)
            st = ctypes.c_void_p()
            with torch.cuda.device(device):
                check(lib.nmv_ar_create(ctypes.byref(st), rank_in_group, world_size, max_bytes, buf))
            self.state, handle = st, buf.raw
        except Exception as e:  # allocation / IPC export refused
            ok, self.disabled_reason = False, f"create: {e}"
        gathered: List = [None] * world_size
        dist.all_gather_object(gathered, (ok, handle, socket.gethostname(), device.index), group=cpu_group)
        if not all(g[0] for g in gathered):
            self.disabled_reason = self.disabled_reason or "a peer could not create its buffer"
            return self._close()
        if len({g[2] for g in gathered}) != 1:
            self.disabled_reason = "ranks on different hosts"
            return self._close()
        try:
            # the kernels read the peers' buffers directly: refuse unless the runtime reports peer access
            # for every other device (ranks that share one device -- the rehearsal -- need none)
            for g in gathered:
                if g[3] is not None and device.index is not None and g[3] != device.index \
                        and not torch.cuda.can_device_access_peer(device.index, g[3]):
                    raise RuntimeError(f"device {device.index} cannot access peer device {g[3]}")
            with torch.cuda.device(device):
                check(lib.nmv_ar_open(self.state, b"".join(g[1] for g in gathered)))
            # bound of a flag wait (default 2 s): ranks that time-share one GPU (rehearsals) need a longer one
            if os.environ.get("NMV_CUSTOM_AR_TIMEOUT_MS"):
                check(lib.nmv_ar_set_timeout_ms(self.state, int(os.environ["NMV_CUSTOM_AR_TIMEOUT_MS"])))
        except Exception as e:
            ok, self.disabled_reason = False, f"open: {e}"
        if not self._all_agree(ok):
            self.disabled_reason = self.disabled_reason or "a peer could not map the buffers"
            return self._close()
        self.enabled = True  # for the self-test calls
        good = self._self_test()
        if not self._all_agree(good):
            self.disabled_reason = self.disabled_reason or "self-test failed on a peer"
            self.enabled = False
            return self._close()

    # ------------------------------------------------------------------
    def _all_agree(self, flag: bool) -> bool:
        flags: List = [None] * self.world
        dist.all_gather_object(flags, bool(flag), group=self.cpu_group)
        return all(flags)

    def _close(self) -> None:
        if self.state is not None:
            with torch.cuda.device(self.device):
                _lib.load().nmv_ar_destroy(self.state)
            self.state = None
        self.enabled = False

    def close(self) -> None:
        self._close()

    def _self_test(self) -> bool:
        """bit-exact against the fp32 rank-order sum computed on the CPU, several sizes and rounds.
        Every rank walks the same sequence of CPU collectives whatever it finds (a rank that bailed
        out early would pair its next collective with its peers' current one); a failure only clears
        the flag that the caller then shares."""
        good = True
        lib = _lib.load()
        # (message size, algorithm): 0 = the reference's size rule, 1 / 2 = one-shot / two-shot forced
        cases = [(8, 0), (4096, 0), (64 * 4096, 0), (64 * 4096 + 8, 0), (1 << 20, 0),
                 (8, 2), (64 * 4096 + 8, 2), (1 << 20, 1), (1 << 20, 2)]
        for rnd, (numel, algo) in enumerate(cases):
            lib.nmv_ar_set_algo(self.state, algo)
            for dtype in (torch.bfloat16, torch.float16):
                g = torch.Generator().manual_seed(1000 * rnd + self.rank)
                x = torch.randn(numel, generator=g).to(dtype)
                parts: List = [None] * self.world
                dist.all_gather_object(parts, x, group=self.cpu_group)
                try:
                    ref = torch.zeros(numel, dtype=torch.float32)
                    for p in parts:
                        ref += p.float()
                    got = self.all_reduce(x.to(self.device))
                    torch.cuda.synchronize(self.device)
                    if not torch.equal(got.cpu().view(torch.int16), ref.to(dtype).view(torch.int16)):
                        good = False
                        self.disabled_reason = self.disabled_reason or \
                            f"self-test mismatch (numel={numel}, {dtype}, algo={algo})"
                except Exception as e:
                    good = False
                    self.disabled_reason = self.disabled_reason or f"self-test: {e}"
        try:
            lib.nmv_ar_set_algo(self.state, 0)
            if not self._gather_ok():
                good = False
                self.disabled_reason = self.disabled_reason or "self-test: all_gather mismatch"
            for algo in (1, 2, 0):   # the fused variants in both forms; ends on the size rule
                lib.nmv_ar_set_algo(self.state, algo)
                if not self._fused_ok():
                    good = False
                    self.disabled_reason = self.disabled_reason or \
                        f"self-test: fused all-reduce + norm mismatch (algo={algo})"
            if _lib.load().nmv_ar_error(self.state):
                good = False
                self.disabled_reason = self.disabled_reason or "self-test: a flag wait timed out"
        except Exception as e:
            good = False
            self.disabled_reason = self.disabled_reason or f"self-test: {e}"
        return good

    # ------------------------------------------------------------------
    def set_algo(self, algo: int) -> None:
        """0 = the reference's size rule (custom_all_reduce.cuh:442-451), 1 = one-shot, 2 = two-shot;
        every rank must make the same call"""
        check(_lib.load().nmv_ar_set_algo(self.state, algo))

    def set_timeout_ms(self, ms: int) -> None:
        check(_lib.load().nmv_ar_set_timeout_ms(self.state, ms))

    def is_two_shot(self, nbytes: int) -> bool:
        return bool(_lib.load().nmv_ar_is_two_shot(self.state, nbytes))

    def local_error(self) -> bool:
        """True when a flag wait ran out on THIS rank (synchronises the device)"""
        if self.state is None:
            return False
        with torch.cuda.device(self.device):
            return bool(_lib.load().nmv_ar_error(self.state))

    def check_error(self, collective: bool = True) -> None:
        """raise CustomAllReduceError when a flag wait timed out -- on any rank of the group when
        `collective` (every rank must then make this call), else on this rank only"""
        bad = self.local_error()
        if collective:
            bad = not self._all_agree(not bad)
        if bad:
            self.enabled = False
            raise CustomAllReduceError(
                "custom all-reduce: a peer did not arrive within the flag-wait bound; the affected "
                "outputs are NaN and the communicator is out of step -- results are invalid")

    def should_use(self, t: torch.Tensor) -> bool:
        nbytes = t.numel() * t.element_size()
        return (self.enabled and t.is_cuda and t.dtype in (torch.float16, torch.bfloat16) and t.is_contiguous()
                and 0 < nbytes <= self.max_bytes and nbytes % 16 == 0 and t.data_ptr() % 16 == 0)

    def all_reduce(self, t: torch.Tensor) -> torch.Tensor:
        """out-of-place sum over the group; the same bits on every rank"""
        out = torch.empty_like(t)
        check(_lib.load().nmv_ar_all_reduce(self.state, ptr(t), ptr(out), t.numel(), dtype_code(t.dtype),
                                            stream_of(t)))
        return out


    def all_reduce_partial(self, slab: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
        """all-reduce of round(sum of the fp32 split-K slabs [S, T, N]) -> [T, N] in `dtype`"""
        assert self.enabled and slab.dtype == torch.float32 and slab.is_contiguous() and slab.dim() == 3
        out = torch.empty(slab.shape[1:], dtype=dtype, device=slab.device)
        check(_lib.load().nmv_ar_all_reduce_partial(self.state, ptr(slab), slab.shape[0], ptr(out), out.numel(),
                                                    dtype_code(dtype), stream_of(slab)))
        return out

    def all_reduce_add_rms_norm(self, x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor,
                                epsilon: float) -> torch.Tensor:
        """x: [T, H] in the model dtype, or the fp32 split-K slabs [S, T, H]; residual += all_reduce(x)
        in place; returns rms_norm(residual) * weight -- one launch"""
        slabs = x.dtype == torch.float32
        assert self.enabled and x.is_contiguous() and residual.is_contiguous() and x.dim() == (3 if slabs else 2)
        rows, hidden = x.shape[-2], x.shape[-1]
        assert residual.numel() == rows * hidden
        out = torch.empty_like(residual)
        check(_lib.load().nmv_ar_all_reduce_add_rms_norm(
            self.state, None if slabs else ptr(x), ptr(x) if slabs else None, x.shape[0] if slabs else 0,
            ptr(residual), ptr(weight), ptr(out), epsilon, rows, hidden, dtype_code(residual.dtype),
            stream_of(residual)))
        return out

    def can_reduce(self, numel: int) -> bool:
        return self.enabled and 0 < numel * 2 <= self.max_bytes and (numel * 2) % 16 == 0

    def all_gather_record(self, t: torch.Tensor) -> torch.Tensor:
        """[world, *t.shape]: every rank's small contiguous record (nbytes % 16 == 0), rank order"""
        nbytes = t.numel() * t.element_size()
        assert self.enabled and t.is_contiguous() and nbytes % 16 == 0 and nbytes <= self.max_bytes
        out = torch.empty((self.world, ) + tuple(t.shape), dtype=t.dtype, device=t.device)
        check(_lib.load().nmv_ar_all_gather(self.state, ptr(t), ptr(out), nbytes, stream_of(t)))
        return out

    # also part of the self-test
    def _gather_ok(self) -> bool:
        x = torch.arange(64, dtype=torch.float32, device=self.device) + 1000.0 * self.rank
        got = self.all_gather_record(x).cpu()
        ref = torch.stack([torch.arange(64, dtype=torch.float32) + 1000.0 * r for r in range(self.world)])
        return torch.equal(got, ref)


    def _fused_ok(self) -> bool:
        """the slab-consuming and the norm-fusing variants against the (just verified) plain all-reduce
        followed by the stand-alone fused_add_rms_norm: bit for bit"""
        from .. import _custom_ops as ops
        ok_all = True
        for rows, hidden in ((64, 4096), (3, 8192)):
            g = torch.Generator().manual_seed(99 + self.rank)
            x = torch.randn((rows, hidden), generator=g).to(torch.bfloat16).to(self.device)
            g2 = torch.Generator().manual_seed(5)
            res = torch.randn((rows, hidden), generator=g2).to(torch.bfloat16).to(self.device)
            w = (1 + 0.1 * torch.randn((hidden, ), generator=g2)).to(torch.bfloat16).to(self.device)
            ref = self.all_reduce(x)
            slabs = torch.stack([x.float() * 0.5, x.float() * 0.5])
            good = True
            if not torch.equal(self.all_reduce_partial(slabs, torch.bfloat16), ref):
                good = False
            ref_res = res.clone()
            ops.fused_add_rms_norm(ref, ref_res, w, 1e-5)
            for inp in (x, slabs):   # every rank issues every launch, whatever it has seen so far
                got_res = res.clone()
                got = self.all_reduce_add_rms_norm(inp, got_res, w, 1e-5)
                if not (torch.equal(got, ref) and torch.equal(got_res, ref_res)):
                    good = False
            if not good:
                ok_all = False
        torch.cuda.synchronize(self.device)
        return ok_all


def maybe_create(cpu_group, rank_in_group: int, world_size: int, device: torch.device,
                 backend: str) -> Optional[CustomAllReduce]:
    """NMV_CUSTOM_ALLREDUCE: "1" (default) = on for RCCL groups of 2..8 GPUs, "0" = off,
    "force" = also over a gloo group (the single-GPU rehearsal of the tests)"""
    mode = os.environ.get("NMV_CUSTOM_ALLREDUCE", "1")
    if mode == "0" or world_size < 2 or world_size > 8 or device.type != "cuda":
        return None
    if backend != "nccl" and mode != "force":
        return None
    car = CustomAllReduce(cpu_group, rank_in_group, world_size, device)
    if not car.enabled and rank_in_group == 0:
        import sys
        print(f"[custom_all_reduce] not used ({car.disabled_reason}); collectives stay on the process group",
              file=sys.stderr)
    return car if car.enabled else None
