"""CPU: the drop-in boundary, op by op.  Every in-scope op of the reference's csrc/torch_bindings.cpp (schema
strings, incl. the ones torch infers from the C++ signatures of csrc/ops.h) and every in-scope function of
vllm/_custom_ops.py (name + parameter names, in order) against this package's registrations.  The reference side
is the committed snapshot tests/golden/reference_op_surface.json (tools/make_op_surface.py); where
/root/reference exists the snapshot is re-derived and must be current."""
import inspect
import json
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SNAP = os.path.join(ROOT, "tests", "golden", "reference_op_surface.json")

# ops of the reference that are NOT on the hot path (out of scope) -- everything else
# that the reference registers must exist here with the identical schema
OUT_OF_SCOPE_OPS = {"aqlm_gemm", "aqlm_dequant", "squeezellm_gemm", "gptq_marlin_24_gemm", "moe_align_block_size",
                    "topk_softmax"}
OUT_OF_SCOPE_FUNCS = OUT_OF_SCOPE_OPS | {"dispatch_bgmv", "dispatch_bgmv_low_level", "is_custom_op_supported",
                                         "hint_on_error"}


def norm(s):
    return re.sub(r"\s+", " ", s).replace("( ", "(").replace(" )", ")").strip()


def snap():
    return json.load(open(SNAP))


def test_snapshot_is_current():
    if not os.path.exists("/root/reference/csrc/torch_bindings.cpp"):
        pytest.skip("reference not on this box: the committed snapshot is what is compared")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_op_surface
    assert make_op_surface.snapshot() == snap(), "run tools/make_op_surface.py"


def test_every_in_scope_schema_is_identical():
    import neural_magic_vllm_amd._torch_bindings as tb
    ours = {ns: {s.split("(")[0]: norm(s) for s in lst} for ns, lst in tb.all_schemas().items()}
    ref = snap()["torch_bindings"]
    for ns in ("_C", "_C_cache_ops", "_C_cuda_utils", "_C_custom_ar"):
        want = {k: norm(v["schema"]) for k, v in ref[ns].items() if k not in OUT_OF_SCOPE_OPS}
        missing = sorted(set(want) - set(ours.get(ns, {})))
        assert not missing, f"{ns}: not registered here: {missing}"
        for name, schema in want.items():
            assert ours[ns][name] == schema, f"{ns}::{name}\n ours: {ours[ns][name]}\n ref:  {schema}"
        extra = sorted(set(ours[ns]) - set(ref[ns]))
        assert not extra, f"{ns}: registered here but not an op of the reference: {extra}"


def test_registered_ops_carry_those_schemas():
    """what torch actually holds for the ops (the strings above are what was asked for)"""
    import torch
    import neural_magic_vllm_amd  # noqa: F401
    ref = snap()["torch_bindings"]
    for ns, ops in ref.items():
        for name, v in ops.items():
            if name in OUT_OF_SCOPE_OPS:
                continue
            op = getattr(getattr(torch.ops, ns), name).default
            got = str(op._schema)
            # torch prints "Tensor(a0!) out" for "Tensor! out": compare names, types and order
            strip = lambda s: re.sub(r"\([\w$]+!?(?: -> \w*)?\)", "", s.replace("Tensor[]!", "Tensor[]").replace("Tensor!", "Tensor"))  # noqa: E731
            want = norm(v["schema"])
            assert strip(got).split("(", 1)[1].replace(" ", "") == strip(want).split("(", 1)[1].replace(" ", ""), \
                f"{ns}::{name}\n torch: {got}\n ref:   {want}"


def test_custom_ops_functions_match():
    from neural_magic_vllm_amd import _custom_ops as ops
    ref = snap()["custom_ops"]
    for name, params in ref.items():
        if name in OUT_OF_SCOPE_FUNCS:
            continue
        fn = getattr(ops, name, None)
        assert fn is not None, f"_custom_ops.{name} is missing"
        got = [p for p in inspect.signature(fn).parameters]
        assert got == params, f"_custom_ops.{name}: {got} != reference {params}"
