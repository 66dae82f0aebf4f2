"""CPU: the `_C` extension (csrc/torch_bindings.cpp) is built, imports, registers all four namespaces with the
reference's schemas (tests/test_op_surface.py then runs against ITS registrations -- the default binding), and is
linked against the C ABI version this package expects."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_extension_is_the_default_binding():
    so = glob.glob(os.path.join(ROOT, "neural_magic_vllm_amd", "_C*.so"))
    assert len(so) == 1, "run __graft_entry__.build() (neural_magic_vllm_amd/csrc/setup_C.py)"
    code = ("import neural_magic_vllm_amd, torch\n"
            "from neural_magic_vllm_amd import _torch_bindings as tb, _lib, _C\n"
            "assert tb.binding == 'cpp', tb.binding\n"
            "assert _C.abi_version() == _lib.ABI_VERSION\n"
            "for ns, lst in tb.all_schemas().items():\n"
            "    for s in lst:\n"
            "        assert hasattr(getattr(torch.ops, ns), s.split('(')[0]), (ns, s)\n"
            "print('ok')\n")
    env = {k: v for k, v in os.environ.items() if k not in ("NMV_BINDING", "NMV_HIP_LIB")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=env, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_python_binding_stays_selectable():
    code = ("import neural_magic_vllm_amd\nfrom neural_magic_vllm_amd import _torch_bindings as tb\n"
            "assert tb.binding == 'python', tb.binding\nprint('ok')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT,
                       env=dict(os.environ, NMV_BINDING="python"), timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
