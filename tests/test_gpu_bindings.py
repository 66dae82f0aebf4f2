"""GPU: the two registrations of the op boundary -- the `_C` TORCH_LIBRARY extension (csrc/torch_bindings.cpp, the form
`import vllm._C` has in the reference: csrc/torch_bindings.cpp:18-294, csrc/registration.h:17-22) and the torch.library
registration of _torch_bindings.py -- produce the same bits for the same calls, raise for the same misuse, and the
extension alone (torch.ops.load_library, no package import) serves an op."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _probe(binding, path):
    env = dict(os.environ, NMV_BINDING=binding, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    subprocess.run([sys.executable, os.path.join(ROOT, "tests", "binding_probe.py"), path], check=True, env=env,
                   cwd=ROOT, timeout=600)
    return torch.load(path)


def test_cpp_and_python_bindings_agree_bit_for_bit(tmp_path):
    cpp = _probe("cpp", str(tmp_path / "cpp.pt"))
    py = _probe("python", str(tmp_path / "py.pt"))
    assert cpp.pop("binding") == "cpp" and py.pop("binding") == "python"
    assert cpp.keys() == py.keys() and len(cpp) >= 14
    for name in cpp:
        a, b = cpp[name], py[name]
        if isinstance(a, torch.Tensor):
            assert a.dtype == b.dtype and a.shape == b.shape, name
            assert torch.equal(a.view(torch.uint8) if a.dtype.is_floating_point else a,
                               b.view(torch.uint8) if b.dtype.is_floating_point else b), name
        else:
            assert a == b, (name, a, b)
    assert cpp["cpu_call"] in ("NotImplementedError", "RuntimeError") and cpp["bad_bits"] is True


def test_extension_alone_serves_an_op(tmp_path):
    """what an unmodified `vllm/_custom_ops.py` does: load the shared object, call torch.ops._C.*"""
    code = f"""
import glob, torch
so, = glob.glob({os.path.join(ROOT, 'neural_magic_vllm_amd', '_C*.so')!r})
torch.ops.load_library(so)
x = torch.randn(5, 256, dtype=torch.float16, device='cuda')
w = torch.ones(256, dtype=torch.float16, device='cuda')
y = torch.empty_like(x)
torch.ops._C.rms_norm(y, x, w, 1e-6)
ref = x.float() * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + 1e-6)
assert (y.float() - ref).abs().max().item() < 2e-3
assert torch.ops._C_cuda_utils.get_max_shared_memory_per_block_device_attribute(0) > 0
print('ok')
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
