import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
import oracle
from test_oracle_golden_w8a8 import load_mm, MM
from neural_magic_vllm_amd import _custom_ops as ops
dev = torch.device("cuda:0")
for name in MM:
    c = load_mm(name)
    if c["kind"] != "int8": continue
    a = c["a"].to(dev); b = c["b"].t().contiguous().to(dev).t()
    out = ops.cutlass_scaled_mm(a, b, c["scale_a"].to(dev), c["scale_b"].to(dev), c["dtype"], None if c["bias"] is None else c["bias"].to(dev)).cpu()
    ref = oracle.scaled_mm(c["a"], c["b"], c["scale_a"], c["scale_b"], c["dtype"], c["bias"])
    acc = (c["a"].double() @ c["b"].double())
    sa, sb = c["scale_a"].float(), c["scale_b"].float()
    t1 = (sa * (sb * acc.float()))          # fp32 two roundings
    exact = sa.double() * sb.double() * acc
    if c["bias"] is not None:
        exact = exact + c["bias"].double()
    bad = (out.view(torch.int16) != ref.view(torch.int16))
    print(name, "mismatch", int(bad.sum()), "of", bad.numel(), "shape sa", tuple(sa.shape), "sb", tuple(sb.shape))
    if bad.any():
        idx = bad.nonzero()[:5]
        for i, j in idx.tolist():
            print("   ", i, j, "hip", float(out[i, j]), "oracle", float(ref[i, j]), "exact", float(exact[i, j]), "acc", float(acc[i, j]), "sa", float(sa[i if sa.shape[0] > 1 else 0, 0]), "sb", float(sb[0, j if sb.shape[1] > 1 else 0]))
