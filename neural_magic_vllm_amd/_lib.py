"""ctypes binding of libnmvllm_hip.so -- the C ABI declared in include/nmvllm_hip.h.

This is the only place the product touches native code.  There is NO fallback: if the shared
library is missing or a call fails, an exception is raised (the analogue of the reference's
`import vllm._C` failure, vllm/_custom_ops.py:11-14, except that here it is fatal).
"""
import ctypes
import os
from ctypes import c_float, c_int, c_int32, c_int64, c_void_p
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# NMV_HIP_LIB: kernel-development override (A/B builds of the same C ABI); never a CPU fallback
LIB_PATH = os.environ.get("NMV_HIP_LIB") or os.path.join(_HERE, "libnmvllm_hip.so")

# enums of include/nmvllm_hip.h
NMV_F16, NMV_BF16, NMV_F32 = 0, 1, 2
NMV_KV_AUTO, NMV_KV_FP8_E4M3 = 0, 1
NMV_I8, NMV_FP8_E4M3 = 0, 1

_DTYPE = {torch.float16: NMV_F16, torch.bfloat16: NMV_BF16, torch.float32: NMV_F32}


class NmvError(RuntimeError):
    """A C-ABI call returned non-zero (TORCH_CHECK -> RuntimeError in the reference)."""


_lib: Optional[ctypes.CDLL] = None
ABI_VERSION = 8   # nmv_abi_version() of the library this table describes (csrc/capi_common.hip)

_P = c_void_p
_I = c_int
_L = c_int64
_F = c_float

# name -> (restype, argtypes); must list every symbol include/nmvllm_hip.h declares
# (tests/test_capi_symbols.py parses the header and checks this table and the .so against it).
SIGNATURES = {
    "nmv_last_error": (ctypes.c_char_p, []),
    "nmv_abi_version": (_I, []),
    "nmv_reshape_and_cache": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _L, _L, _I, _I, _F, _P]),
    "nmv_reshape_and_cache_flash": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _L, _L, _L, _I, _P]),
    "nmv_copy_blocks": (_I, [_P, _P, _P, _I, _I, _L, _I, _P]),
    "nmv_swap_blocks": (_I, [_P, _P, _P, _I, _L, _I, _P]),
    "nmv_convert_fp8": (_I, [_P, _P, _L, _L, _I, _I, _F, _P]),
    "nmv_paged_attention_v1": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _I, _I, _I, _P,
                                    _L, _L, _L, _I, _I, _F, _I, _I, _I, _I, _I, _P]),
    "nmv_paged_attention_v2": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _I,
                                    _I, _I, _P, _L, _L, _L, _I, _I, _F, _I, _I, _I, _I, _I, _P]),
    "nmv_rms_norm": (_I, [_P, _P, _P, _F, _I, _I, _I, _P]),
    "nmv_fused_add_rms_norm": (_I, [_P, _P, _P, _F, _I, _I, _I, _P]),
    "nmv_rotary_embedding": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _L, _L, _P, _I, _I, _P]),
    "nmv_batched_rotary_embedding": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _L, _L, _P, _I, _P, _I,
                                          _P]),
    "nmv_act_and_mul": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "nmv_rotary_embedding_and_cache": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _L, _L, _L, _P, _I, _P, _P,
                                            _P, _I, _I, _I, _F, _P]),
    "nmv_gptq_marlin_gemm_silu_mul": (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _P]),
    "nmv_gptq_marlin_gemm_partial_splits": (_I, [_I, _I, _I, _I]),
    "nmv_gptq_marlin_gemm_partial": (_I, [_P, _L, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "nmv_fused_add_rms_norm_partial": (_I, [_P, _P, _I, _P, _P, _F, _I, _I, _I, _P]),
    "nmv_fused_add_rms_norm_partial16": (_I, [_P, _P, _I, _P, _P, _F, _I, _I, _I, _P]),
    "nmv_rotary_embedding_and_cache_partial": (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _I, _I,
                                                    _F, _P]),
    "nmv_rotary_embedding_and_cache_partial16": (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _I, _I,
                                                    _F, _P]),
    "nmv_ar_handle_bytes": (_I, []),
    "nmv_ar_create": (_I, [_P, _I, _I, _L, _P]),
    "nmv_ar_open": (_I, [_P, _P]),
    "nmv_ar_all_reduce": (_I, [_P, _P, _P, _L, _I, _P]),
    "nmv_ar_all_reduce_partial": (_I, [_P, _P, _I, _P, _L, _I, _P]),
    "nmv_ar_all_reduce_add_rms_norm": (_I, [_P, _P, _P, _I, _P, _P, _P, _F, _I, _I, _I, _P]),
    "nmv_ar_all_gather": (_I, [_P, _P, _P, _L, _P]),
    "nmv_ar_error": (_I, [_P]),
    "nmv_ar_set_algo": (_I, [_P, _I]),
    "nmv_ar_is_two_shot": (_I, [_P, _L]),
    "nmv_ar_set_timeout_ms": (_I, [_P, _L]),
    "nmv_w4_native_repack": (_I, [_P, _P, _P, _I, _I, _P]),
    "nmv_w4_native_gemm_splits": (_I, [_I, _I, _I, _I]),
    "nmv_w4_native_gemm": (_I, [_P, _P, _P, _P, _P, _L, _P, _L, _I, _I, _I, _I, _I, _I, _P]),
    "nmv_w4_ring_timeouts": (_I, []),
    "nmv_prefetch_l3": (_I, [_P, _L, _I, _P]),
    "nmv_w4_native_prefill_plan": (_I, [_I, _I, _I]),
    "nmv_w4_native_gemm_slab16": (_I, [_I, _I, _I]),
    "nmv_car_meta_size": (_L, []),
    "nmv_car_meta_alloc": (_I, [_L, _P, _P]),
    "nmv_car_meta_free": (_I, [_P]),
    "nmv_car_init": (_I, [_P, _P, _P, _L, _P, _P, _I, _I, _I]),
    "nmv_car_register_buffer": (_I, [_P, _P, _P, _P]),
    "nmv_car_all_reduce": (_I, [_P, _P, _P, _L, _I, _P]),
    "nmv_car_graph_buffer_count": (_I, [_P]),
    "nmv_car_get_graph_buffer_ipc_meta": (_I, [_P, _P, _P]),
    "nmv_car_register_graph_buffers": (_I, [_P, _P, _P]),
    "nmv_car_set_algo": (_I, [_P, _I]),
    "nmv_car_error": (_I, [_P]),
    "nmv_car_dispose": (_I, [_P]),
    "nmv_greedy_record_elems": (_I, [_I]),
    "nmv_greedy_sample_shard": (_I, [_P, _P, _L, _I, _I, _I, _I, _P, _L, _P]),
    "nmv_greedy_sample_finish": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _I, _I, _P]),
    "nmv_ar_destroy": (_I, [_P]),
    "nmv_paged_attention_v1_rope_partial": (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _I, _I,
                                                 _I, _L, _L, _I, _I, _F, _P]),
    "nmv_paged_attention_v2_rope_partial": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P,
                                                 _P, _I, _I, _I, _L, _L, _I, _I, _F, _P]),
    "nmv_greedy_sample_scratch_bytes": (_L, [_I]),
    "nmv_greedy_sample_advance": (_I, [_P, _P, _L, _I, _I, _I, _P, _L, _P, _P, _P, _P, _P, _I, _I, _P]),
    "nmv_rms_norm_dynamic_int8_quant": (_I, [_P, _P, _P, _P, _P, _F, _I, _I, _I, _P]),
    "nmv_silu_and_mul_dynamic_int8_quant": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "nmv_activation": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "nmv_gptq_marlin_repack": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "nmv_gptq_marlin_gemm_scratch_bytes": (_L, [_I, _I, _I, _I]),
    "nmv_gptq_marlin_gemm": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _P, _L, _I, _I, _I, _I, _I, _I,
                                  _I, _P]),
    "nmv_marlin_gemm": (_I, [_P, _P, _P, _P, _P, _L, _P, _L, _I, _I, _I, _I, _I, _P]),
    "nmv_fp8_marlin_gemm_scratch_bytes": (_L, [_I, _I, _I]),
    "nmv_fp8_marlin_gemm": (_I, [_P, _P, _P, _P, _P, _L, _P, _L, _I, _I, _I, _I, _I, _I, _P]),
    "nmv_gptq_gemm": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _L, _P]),
    "nmv_wq_gemm_scratch_bytes": (_L, [_I, _I, _I]),
    "nmv_marlin_zp_gemm": (_I, [_P, _P, _P, _P, _P, _P, _L, _P, _L, _I, _I, _I, _I, _I, _P]),
    "nmv_awq_marlin_repack": (_I, [_P, _P, _I, _I, _P]),
    "nmv_prefill_attention_supported": (_I, [_I]),
    "nmv_prefix_prefill_attention": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F,
                                          _L, _L, _L, _L, _P, _I, _I, _P]),
    "nmv_prefill_attention": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _L, _L, _L, _P, _I, _I, _P]),
    "nmv_gptq_shuffle": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "nmv_awq_gemm": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _L, _P]),
    "nmv_awq_dequantize": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "nmv_scaled_int8_quant": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "nmv_scaled_fp8_quant": (_I, [_P, _P, _P, _L, _I, _I, _P]),
    "nmv_scaled_mm": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _L, _L, _L, _I, _I, _I, _I, _P, _L, _P]),
    "nmv_scaled_mm_scratch_bytes": (_L, [_I, _I, _I]),
    "nmv_cutlass_scaled_mm_supports_fp8": (_I, [_L]),
    "nmv_get_device_attribute": (_L, [_L, _L]),
    "nmv_get_max_shared_memory_per_block_device_attribute": (_L, [_L]),
}


def load() -> ctypes.CDLL:
    """Load the HIP library; raise loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the gfx950 HIP library has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (or `make -C "
            "neural_magic_vllm_amd/csrc`). There is no CPU/eager fallback for these ops.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    # a stale build that still exports every symbol would be called with shifted arguments: refuse it
    got = lib.nmv_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} has C ABI version {got}, this package needs {ABI_VERSION}: rebuild it "
                          "(`make -C neural_magic_vllm_amd/csrc`)")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().nmv_last_error().decode("utf-8", "replace")
        raise NmvError(msg or f"{what} failed with code {rc}")


def dtype_code(dt: torch.dtype) -> int:
    try:
        return _DTYPE[dt]
    except KeyError:
        raise NmvError(f"unsupported data type {dt}") from None


def kv_dtype_code(kv_cache_dtype: str) -> int:
    # reference: DISPATCH_BY_KV_CACHE_DTYPE, csrc/quantization/fp8/nvidia/quant_utils.cuh:545-571
    if kv_cache_dtype == "auto":
        return NMV_KV_AUTO
    if kv_cache_dtype in ("fp8", "fp8_e4m3"):
        return NMV_KV_FP8_E4M3
    raise NmvError(f"Unsupported data type of kv cache: {kv_cache_dtype}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def stream_of(t: torch.Tensor) -> int:
    """Current HIP stream of the tensor's device (the capturing stream during graph capture)."""
    return torch.cuda.current_stream(t.device).cuda_stream


class device_guard:
    """at::cuda::OptionalCUDAGuard analogue: only switches when the tensor is elsewhere."""

    def __init__(self, t: torch.Tensor):
        if t.device.type != "cuda":   # a host pointer handed to a kernel is a GPU fault, not an exception
            raise NmvError(f"expected a tensor on the GPU, got one on {t.device} (there is no CPU path)")
        self.idx = t.device.index
        self.prev = None

    def __enter__(self):
        if self.idx is not None:
            cur = torch.cuda.current_device()
            if cur != self.idx:
                self.prev = cur
                torch.cuda.set_device(self.idx)
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)
        return False
