"""Build recipe for oracle/_ref: the REFERENCE's own CPU kernels, compiled from where they lie.

TEST INFRASTRUCTURE ONLY.  Compiles /root/reference/csrc/cpu/{activation,attention,cache,layernorm,
pos_encoding,torch_bindings}.cpp with g++ (through torch.utils.cpp_extension, i.e. the same
flags the reference's cmake/cpu_extension.cmake:58-63 uses for AVX512 without avx512_bf16) into
oracle/_ref/_C_ref.so.  Nothing is copied: sources are read in place, only the object/.so land
under oracle/_ref/ (git-ignored, but it travels to the GPU box with the snapshot).

TORCH_EXTENSION_NAME is set to `_C_ref`, so the ops register as torch.ops._C_ref.* and
torch.ops._C_ref_cache_ops.* (csrc/cpu/torch_bindings.cpp:7,87 expand the macro) and do not
collide with the product's torch.ops._C namespace.

Runs only where /root/reference exists (this container).  On the GPU box the prebuilt .so is
loaded if present (load_ref()), and everything that needs it is skipped otherwise.
"""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = "/root/reference"
OUT_DIR = os.path.join(_HERE, "_ref")
SO = os.path.join(OUT_DIR, "_C_ref.so")
_SRCS = ["activation.cpp", "attention.cpp", "cache.cpp", "layernorm.cpp", "pos_encoding.cpp",
         "torch_bindings.cpp"]


def have_reference() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "csrc", "cpu"))


def build(verbose: bool = False) -> str:
    if os.path.exists(SO):
        return SO
    if not have_reference():
        raise RuntimeError("reference sources not present; oracle/_ref cannot be built here")
    from torch.utils.cpp_extension import load
    os.makedirs(OUT_DIR, exist_ok=True)
    csrc = os.path.join(REF_ROOT, "csrc")
    load(name="_C_ref",
         sources=[os.path.join(csrc, "cpu", s) for s in _SRCS],
         extra_include_paths=[csrc],
         extra_cflags=["-O2", "-fopenmp", "-DVLLM_CPU_EXTENSION", "-mavx512f", "-mavx512vl",
                       "-mavx512bw", "-mavx512dq", "-std=c++17",
                       "-DTORCH_EXTENSION_NAME=_C_ref"],
         extra_ldflags=["-fopenmp"],
         build_directory=OUT_DIR,
         is_python_module=False,
         verbose=verbose)
    assert os.path.exists(SO), "expected " + SO
    return SO


_loaded = False


def load_ref() -> bool:
    """torch.ops.load_library(oracle/_ref/_C_ref.so) if it exists (and the host has AVX512)."""
    global _loaded
    if _loaded:
        return True
    if not os.path.exists(SO):
        return False
    try:
        with open("/proc/cpuinfo") as f:
            if "avx512bw" not in f.read():
                return False
    except OSError:
        return False
    import torch
    torch.ops.load_library(SO)
    _loaded = True
    return True


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv))
