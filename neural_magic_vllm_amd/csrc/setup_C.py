"""Builds `neural_magic_vllm_amd/_C.*.so`, the TORCH_LIBRARY form of the boundary (csrc/torch_bindings.cpp), with
torch.utils.cpp_extension.  Host code only (no device code in this extension): the C++ compiler, PyTorch-ROCm's
headers, and a link against libnmvllm_hip.so beside it (rpath $ORIGIN), which `make -C neural_magic_vllm_amd/csrc`
must have produced first.  Run from the repository root:

    python neural_magic_vllm_amd/csrc/setup_C.py build_ext --inplace
"""
import os

from setuptools import setup
from torch.utils.cpp_extension import ROCM_HOME, BuildExtension, CppExtension

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
rocm = ROCM_HOME or "/opt/rocm"

assert os.path.exists(os.path.join(PKG, "libnmvllm_hip.so")), "build libnmvllm_hip.so first (make -C neural_magic_vllm_amd/csrc)"

ext = CppExtension(
    name="neural_magic_vllm_amd._C",
    sources=[os.path.relpath(os.path.join(HERE, "torch_bindings.cpp"), ROOT)],
    include_dirs=[os.path.join(ROOT, "include"), os.path.join(rocm, "include")],
    define_macros=[("__HIP_PLATFORM_AMD__", "1"), ("USE_ROCM", "1")],
    library_dirs=[PKG, os.path.join(rocm, "lib")],
    libraries=["c10_hip", "torch_hip", "amdhip64", ":libnmvllm_hip.so"],
    extra_compile_args=["-O2", "-std=c++17", "-Wno-narrowing"],
    extra_link_args=["-Wl,-rpath,$ORIGIN"],
)

setup(name="neural_magic_vllm_amd_C", version="0.1.0", ext_modules=[ext],
      cmdclass={"build_ext": BuildExtension.with_options(use_ninja=True)},
      script_args=None, options={"build": {"build_base": os.path.join(ROOT, "build", "torch_ext")}})
