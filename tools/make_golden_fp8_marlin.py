"""Pin `fp8_marlin_gemm` to the reference's own test recipe (tests/kernels/test_marlin_gemm.py:238-304).  Runs only in the
build container.

The reference holds no vectors for this op, but its test IS a recipe: random activations and weights, the weights
quantised per tensor to fp8-e4m3 (`ops.scaled_fp8_quant`), packed four bytes per int32 (`pack_fp8_to_int32`), repacked to
the 8-bit Marlin tensor (`ops.gptq_marlin_repack`), the per-tensor scale repeated per channel and permuted
(`marlin_permute_scales`), and the kernel's output compared with `torch.matmul(a_input, b_weight)` through
`compute_max_diff(...) < 0.04`.  Here the test function is compiled from the reference's file IN PLACE (ast -> exec,
nothing is copied; "cuda" constants rewritten to "cpu") and run on CPU with its Python helpers as they are
(`pack_fp8_to_int32`, `marlin_permute_scales`, `compute_max_diff`, `rand_data`) and its three CUDA ops bound to this
repo's oracle: `scaled_fp8_quant` and `gptq_marlin_repack` (both already pinned to the reference: fp8_quant.npz,
mq_k128_n128_b8_g64.npz) and `fp8_marlin_gemm` (oracle.fp8_marlin_gemm: unpack -> decode -> scale -> matmul), which the
test's own assert then judges.  What the test drew and expected goes into tests/golden/fp8_marlin_*.npz:
a_input, the Marlin tensor, the permuted scales, `output_ref` (read out of the test's frame) and the sizes.

usage:  python tools/make_golden_fp8_marlin.py
"""
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import helpers  # noqa: E402
import oracle  # noqa: E402
from make_golden_w8a8 import REF, functions_of, save  # noqa: E402


def main():
    sys.modules.setdefault("cpuinfo", types.ModuleType("cpuinfo"))  # optional dep, absent here
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from vllm.model_executor.layers.quantization.gptq_marlin import (GPTQ_MARLIN_MAX_PARALLEL, GPTQ_MARLIN_MIN_THREAD_N,
                                                                     marlin_permute_scales)
    from vllm.model_executor.layers.quantization.utils.marlin_utils import compute_max_diff, pack_fp8_to_int32
    got = {}

    class Workspace:   # MarlinWorkspace allocates on "cuda"; only .scratch is used by the test
        def __init__(self, out_features, min_thread_n, max_parallel):
            self.scratch = torch.zeros(out_features // min_thread_n * max_parallel, dtype=torch.int)

    class Ops:
        @staticmethod
        def scaled_fp8_quant(x, scale=None):
            q, s = oracle.scaled_fp8_quant(x, scale)
            return q.view(torch.float8_e4m3fn) if q.dtype == torch.uint8 else q, s

        @staticmethod
        def gptq_marlin_repack(b_q_weight, perm, size_k, size_n, num_bits):
            return oracle.gptq_marlin_repack(b_q_weight, perm, size_k, size_n, num_bits)

        @staticmethod
        def fp8_marlin_gemm(a, b_q_weight, b_scales, workspace, num_bits, size_m, size_n, size_k):
            got.update(a=a.clone(), q=b_q_weight.clone(), s=b_scales.clone(), m=size_m, n=size_n, k=size_k)
            return oracle.fp8_marlin_gemm(a, b_q_weight, b_scales, size_m, size_n, size_k)

    ns = dict(torch=torch, ops=Ops, pack_fp8_to_int32=pack_fp8_to_int32, marlin_permute_scales=marlin_permute_scales,
              compute_max_diff=compute_max_diff, MarlinWorkspace=Workspace,
              GPTQ_MARLIN_MIN_THREAD_N=GPTQ_MARLIN_MIN_THREAD_N, GPTQ_MARLIN_MAX_PARALLEL=GPTQ_MARLIN_MAX_PARALLEL)
    exec(functions_of(os.path.join(REF, "tests/kernels/test_marlin_gemm.py"), ["rand_data", "test_fp8_marlin_gemm"],
                      rewrite_cuda=True), ns)
    test = ns["test_fp8_marlin_gemm"]

    # the reference grid is k_chunk 128 x n_chunk {64, 128, 256} x MNK_FACTORS x {fp16, bf16} (:31-50); the same
    # function on factors small enough to commit (a fixture holds K * N bytes of weights)
    cases = [(128, 64, (1, 4, 4), torch.bfloat16), (128, 64, (13, 3, 3), torch.float16), (128, 128, (67, 1, 2), torch.bfloat16)]
    for i, (k_chunk, n_chunk, f, dt) in enumerate(cases):
        torch.manual_seed(300 + i)
        got.clear()

        # output_ref is computed after the op returns: read it out of the frame when compute_max_diff is called
        def max_diff(output, output_ref):
            got["ref"] = output_ref.clone()
            got["max_diff"] = float(compute_max_diff(output, output_ref))
            return compute_max_diff(output, output_ref)
        ns["compute_max_diff"] = max_diff
        with mock.patch("torch.cuda.synchronize", lambda *a, **k: None):   # the test's only other touch of the GPU
            test(k_chunk, n_chunk, 8, -1, f, dt)     # asserts max_diff < 0.04 itself (the oracle is what it judges)
        save(f"fp8_marlin_{i}", m=got["m"], n=got["n"], k=got["k"], dtype=str(dt).replace("torch.", ""),
             a=helpers.to_np(got["a"]), marlin_q=got["q"].numpy(), marlin_s=helpers.to_np(got["s"]),
             output_ref=helpers.to_np(got["ref"]), reference_max_diff_of_oracle=np.float32(got["max_diff"]))
        print(f"    case {i}: M N K = {got['m']} {got['n']} {got['k']}  max_diff(oracle) = {got['max_diff']:.4f}")


if __name__ == "__main__":
    main()
