"""Generate tests/golden/bsa_*.npz: block-sparse paged attention pinned to the REFERENCE's own checker.
Runs only in the build container.

The reference's CPU kernels refuse block-sparse attention and its CUDA kernel cannot run here, but the checker of
its kernel test can: `ref_single_query_cached_kv_attention` and `ref_masked_attention` of
tests/kernels/test_blocksparse_attention.py are compiled from the reference's file IN PLACE (ast -> exec; nothing
is copied into this repo) and run on CPU on seeded inputs made with tests/helpers.make_paged_attention_inputs.
The fixture holds the inputs' seed recipe and that function's output; tests/test_oracle_golden.py holds this
repo's Python checker to it and tests/test_gpu_attention.py the HIP kernels (v1 and v2).

usage:  python tools/make_golden_blocksparse.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
GOLD = os.path.join(ROOT, "tests", "golden")

import helpers  # noqa: E402
from make_golden_w8a8 import REF, functions_of  # noqa: E402

# (name, seed, num_seqs, (heads, kv_heads), head_size, block_size, seq_lens, alibi, sparse parameters)
CASES = [
    ("bsa_qslide", 11, 3, (8, 2), 128, 16, [700, 65, 1100], False,
     dict(tp_rank=0, local_blocks=2, vert_stride=4, block_size=64, head_sliding_step=1)),
    ("bsa_kvslide_alibi", 12, 3, (8, 8), 64, 16, [513, 1, 900], True,
     dict(tp_rank=1, local_blocks=3, vert_stride=8, block_size=32, head_sliding_step=-1)),
    ("bsa_homo", 13, 2, (4, 1), 128, 32, [1500, 260], False,
     dict(tp_rank=0, local_blocks=1, vert_stride=3, block_size=64, head_sliding_step=0)),
]


def main():
    from typing import List, Optional, Tuple
    path = os.path.join(REF, "tests", "kernels", "test_blocksparse_attention.py")
    ns = {"torch": torch, "List": List, "Optional": Optional, "Tuple": Tuple}
    exec(functions_of(path, ["ref_masked_attention", "ref_single_query_cached_kv_attention"]), ns)
    ref_fn = ns["ref_single_query_cached_kv_attention"]
    for name, seed, nseq, heads, hs, bs, lens, alibi, sp in CASES:
        inp = helpers.make_paged_attention_inputs(seed, nseq, heads, hs, bs, torch.bfloat16, seq_lens=lens,
                                                  num_blocks=256, use_alibi=alibi)
        q = inp["query"]
        out = torch.empty_like(q)
        ref_fn(out, q, heads[0] // heads[1], inp["key_cache"], inp["value_cache"], inp["block_tables"],
               inp["seq_lens"], inp["scale"], inp["alibi_slopes"], sp["tp_rank"], sp["local_blocks"],
               sp["vert_stride"], sp["block_size"], sp["head_sliding_step"])
        np.savez_compressed(os.path.join(GOLD, name + ".npz"),
                            recipe=np.array([seed, nseq, heads[0], heads[1], hs, bs, int(alibi)], dtype=np.int64),
                            seq_lens=np.array(lens, dtype=np.int64),
                            sparse=np.array([sp["tp_rank"], sp["local_blocks"], sp["vert_stride"], sp["block_size"],
                                             sp["head_sliding_step"]], dtype=np.int64),
                            out=out.float().numpy())
        print(f"  {name}.npz  {os.path.getsize(os.path.join(GOLD, name + '.npz')) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
