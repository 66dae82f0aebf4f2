"""neural_magic_vllm_amd -- MI355X (gfx950) native hot path of nm-vllm 0.5.1.

Scope: paged attention v1/v2, KV-cache ops, the W4A16/W8A16 GPTQ-Marlin
GEMM (+ repack) and the glue ops of a decoder layer, as hand-written HIP kernels behind the
reference's own operator boundary:

    include/nmvllm_hip.h      C ABI of libnmvllm_hip.so (extern "C", raw pointers)
    _torch_bindings.py        torch.ops._C / _C_cache_ops / _C_cuda_utils with the reference schemas
    _custom_ops.py            the vllm._custom_ops Python shim (same names, same arguments)

Importing the package registers the ops (the analogue of `import vllm._C`); the native library
itself is loaded lazily on the first op call and its absence is a hard error.
"""
from . import _torch_bindings

_torch_bindings.register()

__version__ = "0.1.0"
