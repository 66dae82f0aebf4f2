"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

A CPU restatement of the reference's algorithms for the hot path (oracle.c: plain C + OpenMP;
ref_math.py: numpy/torch for the quantisation utilities), used exclusively by tests/,
__graft_entry__.smoke() and the cpu_baseline leg of bench.py as the checker / reported baseline.
The product package (neural_magic_vllm_amd/) never imports this module.

Parity pinning: tests/test_oracle_golden.py checks every function here against golden vectors in
tests/golden/ that tools/make_golden.py produced from the reference itself (its csrc/cpu
kernels compiled into oracle/_ref by oracle/build_ref.py, and its Python quantisation utilities
imported from /root/reference).
"""
import ctypes
import os
import subprocess
from ctypes import c_float, c_int, c_int64, c_void_p
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None

F16, BF16 = 0, 1
_DT = {torch.float16: F16, torch.bfloat16: BF16}


def _host_tag() -> str:
    """-march=native objects must not travel between hosts: key the .so by the CPU's ISA flags."""
    import hashlib
    try:
        with open("/proc/cpuinfo") as f:
            flags = next((ln for ln in f if ln.startswith("flags")), "")
    except OSError:
        flags = ""
    return hashlib.sha1(flags.encode()).hexdigest()[:10]


def so_path() -> str:
    return os.path.join(_HERE, f"liboracle_{_host_tag()}.so")


def build(force: bool = False) -> str:
    """gcc -O3 -march=native -fopenmp oracle.c -> liboracle_<host>.so (a few seconds)."""
    src = os.path.join(_HERE, "oracle.c")
    so = so_path()
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        tmp = so + f".{os.getpid()}.tmp"
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fopenmp", "-fPIC", "-shared",
                               "-std=gnu11", "-o", tmp, src, "-lm"])
        os.replace(tmp, so)
    return so


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _p(t: Optional[torch.Tensor]):
    return None if t is None else c_void_p(t.data_ptr())


def _cpu(t: torch.Tensor) -> torch.Tensor:
    assert t.device.type == "cpu", "the oracle works on CPU tensors"
    return t


def _kv(kv_cache_dtype: str) -> int:
    return 0 if kv_cache_dtype == "auto" else 1


# ---------------------------------------------------------------------------------------------
def reshape_and_cache(key, value, key_cache, value_cache, slot_mapping, kv_cache_dtype="auto",
                      kv_scale=1.0):
    nt, nh, hs = key.shape
    bs = key_cache.shape[3]
    lib().orc_reshape_and_cache(_p(_cpu(key)), _p(value), _p(key_cache), _p(value_cache),
                                _p(slot_mapping), c_int(slot_mapping.numel()), c_int(nh),
                                c_int(hs), c_int(bs), c_int64(key.stride(0)),
                                c_int64(value.stride(0)), c_int(_DT[key.dtype]),
                                c_int(_kv(kv_cache_dtype)), c_float(kv_scale))


def copy_blocks(key_cache, value_cache, block_mapping):
    bm = block_mapping.to(torch.int64).contiguous()
    lib().orc_copy_blocks(_p(key_cache), _p(value_cache), _p(bm), c_int(bm.shape[0]),
                          c_int64(key_cache[0].numel() * key_cache.element_size()))


def paged_attention(query, key_cache, value_cache, num_kv_heads, scale, block_tables, seq_lens,
                    block_size, alibi_slopes=None, kv_cache_dtype="auto", kv_scale=1.0,
                    partition_size=0, max_seq_len=None):
    """v1 (partition_size=0) or v2 (512).  Returns out (and exp_sums, max_logits, tmp_out)."""
    ns, nh, hs = query.shape
    out = torch.empty((ns, nh, hs), dtype=query.dtype)
    if partition_size:
        if max_seq_len is None:
            max_seq_len = int(seq_lens.max())
        mp = (max_seq_len + partition_size - 1) // partition_size
        exp_sums = torch.zeros((ns, nh, mp), dtype=torch.float32)
        max_logits = torch.zeros((ns, nh, mp), dtype=torch.float32)
        tmp_out = torch.zeros((ns, nh, mp, hs), dtype=query.dtype)
    else:
        mp, exp_sums, max_logits, tmp_out = 1, None, None, None
    lib().orc_paged_attention(
        _p(out), _p(exp_sums), _p(max_logits), _p(tmp_out), _p(_cpu(query)), _p(key_cache),
        _p(value_cache), c_int(ns), c_int(nh), c_int(hs), c_int(num_kv_heads), c_float(scale),
        _p(block_tables), _p(seq_lens), c_int(block_size), c_int(block_tables.shape[1]),
        _p(alibi_slopes), c_int64(query.stride(0)), c_int64(key_cache.stride(0)),
        c_int64(key_cache.stride(1)), c_int(_DT[query.dtype]), c_int(_kv(kv_cache_dtype)),
        c_float(kv_scale), c_int(partition_size), c_int(mp))
    if partition_size:
        return out, exp_sums, max_logits, tmp_out
    return out


def gptq_marlin_repack(b_q_weight, perm, size_k, size_n, num_bits):
    pack = 32 // num_bits
    out = torch.empty((size_k // 16, size_n * 16 // pack), dtype=torch.int32)
    has_perm = perm is not None and perm.numel() > 0
    lib().orc_gptq_marlin_repack(_p(_cpu(b_q_weight.contiguous())),
                                 _p(perm.contiguous()) if has_perm else None, _p(out),
                                 c_int(size_k), c_int(size_n), c_int(num_bits))
    return out


def marlin_unpack(marlin_q_w, size_k, size_n, num_bits):
    q = torch.empty((size_k, size_n), dtype=torch.uint8)
    lib().orc_marlin_unpack(_p(_cpu(marlin_q_w.contiguous())), _p(q), c_int(size_k), c_int(size_n),
                            c_int(num_bits))
    return q


def gptq_marlin_gemm(a, b_q_weight, b_scales, g_idx, perm, num_bits, size_m, size_n, size_k):
    c = torch.empty((size_m, size_n), dtype=a.dtype)
    has_ao = g_idx is not None and g_idx.numel() > 0
    lib().orc_gptq_marlin_gemm(_p(c), _p(_cpu(a.contiguous())), _p(b_q_weight.contiguous()),
                               _p(b_scales.contiguous()), _p(g_idx) if has_ao else None,
                               _p(perm) if has_ao else None, c_int(num_bits), c_int(size_m),
                               c_int(size_n), c_int(size_k), c_int(b_scales.shape[0]),
                               c_int(_DT[a.dtype]))
    return c


def rms_norm(input, weight, eps, residual=None):
    """returns out (and updates residual in place when given = fused_add_rms_norm)"""
    out = torch.empty_like(input)
    lib().orc_rms_norm(_p(out), _p(_cpu(input.contiguous())), _p(residual), _p(weight),
                       c_float(eps), c_int(input.numel() // input.shape[-1]),
                       c_int(input.shape[-1]), c_int(_DT[input.dtype]))
    return out


def rotary_embedding(positions, query, key, head_size, cos_sin_cache, is_neox, offsets=None):
    """in place on query/key ([num_tokens, heads*head_size])"""
    nt = query.numel() // query.shape[-1]
    lib().orc_rotary_embedding(_p(positions), _p(_cpu(query)), _p(key), c_int(nt),
                               c_int(query.shape[-1] // head_size),
                               c_int(key.shape[-1] // head_size), c_int(head_size),
                               c_int(cos_sin_cache.shape[1]), c_int64(query.stride(-2)),
                               c_int64(key.stride(-2)), _p(cos_sin_cache), c_int(int(is_neox)),
                               _p(offsets), c_int(_DT[query.dtype]))


def act_and_mul(input, act=0):
    d = input.shape[-1] // 2
    out = torch.empty(input.shape[:-1] + (d, ), dtype=input.dtype)
    lib().orc_act_and_mul(_p(out), _p(_cpu(input.contiguous())),
                          c_int(input.numel() // input.shape[-1]), c_int(d), c_int(act),
                          c_int(_DT[input.dtype]))
    return out


def fp8_decode(t_u8):
    out = torch.empty(t_u8.shape, dtype=torch.float32)
    lib().orc_fp8_decode(_p(_cpu(t_u8.contiguous())), _p(out), c_int64(t_u8.numel()))
    return out


def fp8_encode(t_f32):
    out = torch.empty(t_f32.shape, dtype=torch.uint8)
    lib().orc_fp8_encode(_p(_cpu(t_f32.contiguous().float())), _p(out), c_int64(t_f32.numel()))
    return out


def scaled_int8_quant(input, scale=None):
    """static (scale given, float32[1]) or dynamic per-token; returns (int8 tensor, scales)"""
    out = torch.empty(input.shape, dtype=torch.int8)
    nt = input.numel() // input.shape[-1]
    dynamic = scale is None
    if dynamic:
        scale = torch.empty((nt, 1), dtype=torch.float32)
    lib().orc_scaled_int8_quant(_p(out), _p(_cpu(input.contiguous())), _p(scale), c_int(nt),
                                c_int(input.shape[-1]), c_int(int(dynamic)), c_int(_DT[input.dtype]))
    return out, scale


def scaled_fp8_quant(input, scale=None):
    out = torch.empty(input.shape, dtype=torch.uint8)
    dynamic = scale is None
    if dynamic:
        scale = torch.zeros(1, dtype=torch.float32)
    lib().orc_scaled_fp8_quant(_p(out), _p(_cpu(input.contiguous())), _p(scale),
                               c_int64(input.numel()), c_int(int(dynamic)), c_int(_DT[input.dtype]))
    return out, scale


def fp8_marlin_gemm(a, b_q_weight, b_scales, size_m, size_n, size_k):
    """csrc/quantization/fp8/fp8_marlin.cu:1212-1308 as arithmetic: the 8-bit Marlin tensor holds fp8-e4m3 bytes
    (pack_fp8_to_int32 + gptq_marlin_repack with num_bits = 8), b_scales is the channelwise scale row in
    marlin_permute_scales order; the kernel dequantises byte -> half exactly, multiplies by the channel scale in the
    model dtype and accumulates in fp32.  Composed of pinned pieces: marlin_unpack, fp8_decode."""
    from . import ref_math
    w = fp8_decode(marlin_unpack(b_q_weight, size_k, size_n, 8))                   # [K, N] fp32, exact
    _, single = ref_math.scale_perms()
    inv = torch.empty(len(single), dtype=torch.long)
    inv[torch.tensor(single)] = torch.arange(len(single))
    s = _cpu(b_scales).reshape(-1, len(single))[:, inv].reshape(1, size_n)         # natural column order
    w = (w.to(a.dtype) * s.to(a.dtype)).float()                                    # scale applied in the model dtype
    return (_cpu(a).float() @ w).to(a.dtype)


def scaled_mm(a, b, scale_a, scale_b, out_dtype, bias=None):
    """a [M,K] int8 / float8_e4m3fn, b [K,N] column-major; returns [M,N] in out_dtype"""
    m, k = a.shape
    n = b.shape[1]
    bt = b.t().contiguous()  # [N, K] row-major
    is_fp8 = a.dtype == torch.float8_e4m3fn
    av = a.contiguous().view(torch.uint8) if is_fp8 else a.contiguous().view(torch.uint8)
    btv = bt.view(torch.uint8)
    out = torch.empty((m, n), dtype=out_dtype)
    sa, sb = scale_a.contiguous().float(), scale_b.contiguous().float()
    lib().orc_scaled_mm(_p(out), _p(av), _p(btv), _p(sa), _p(sb), _p(bias), c_int(m), c_int(n),
                        c_int(k), c_int(int(sa.numel() > 1)), c_int(int(sb.numel() > 1)),
                        c_int(int(is_fp8)), c_int(_DT[out_dtype]))
    return out
