"""GPTQ checkpoints on the Marlin kernel: W4A16 / W8A16, symmetric, group 32 / 64 / 128 or channelwise, with or
without act-order -- the method of the headline path (interface and checkpoint contract: reference
vllm/model_executor/layers/quantization/gptq_marlin.py:59-184 config, :192-466 linear method; parameter table pinned
by tests/golden/linear_method_params.json["gptq_marlin"]).

Life of a layer
  create_weights   the GPTQ checkpoint tensors: qweight int32 [K / pack, N], scales [K / g, N], g_idx int32 [K], qzeros
                   (declared on `meta`: symmetric checkpoints carry them, nobody reads them), a ticket array `workspace`
  first call       state REPACK -> READY: act-order sort of g_idx, `gptq_marlin_repack` of the codes into the Marlin
                   interchange tensor (in place, the parameter keeps its storage), `marlin_permute_scales`
  every call       `gptq_marlin_gemm` (csrc/w4a16_stream.hip up to 64 rows, csrc/w4a16_gemm.hip beyond)

Beyond the reference's LinearMethod (used by the decode harness): `apply_partial` leaves the split-K
reduction to the next launch, `apply_silu_mul` folds silu_and_mul into the gate_up GEMM's epilogue on column-interleaved
weights, and for the prevalent format (4-bit symmetric, group 128, no act-order) the first call builds an MFMA-native
tensor of the codes INSTEAD of the Marlin tensor (`nmv_w4_native_repack`; 0.5 byte per weight on the rank): every call of
such a layer goes to `nmv_w4_native_gemm` -- csrc/w4a16_stream.hip / w4a16_ring.hip up to 64 rows, csrc/w4a16_prefill.hip
beyond.  NMV_W4_KEEP_MARLIN=1 keeps both tensors and sends each prompt-sized call to the faster kernel; NMV_W4_NATIVE=0
keeps the Marlin tensor only (the reference's layout, what the op `gptq_marlin_gemm` itself always takes)."""
import enum
import os
from typing import Any, Dict, List, Optional

import torch
from torch.nn.parameter import Parameter

from .... import _custom_ops as ops
from ....platforms import current_platform
from ._schema import Geometry, Require, Slot, build
from .base_config import LinearMethodBase, QuantizationConfig

# geometry of the Marlin interchange format (gptq_marlin.py:14-23)
GPTQ_MARLIN_TILE = 16
GPTQ_MARLIN_MIN_THREAD_N = 64
GPTQ_MARLIN_MIN_THREAD_K = 128
GPTQ_MARLIN_MAX_PARALLEL = 16

GPTQ_MARLIN_SUPPORTED_NUM_BITS = [4, 8]
GPTQ_MARLIN_SUPPORTED_GROUP_SIZES = [-1, 32, 64, 128]
GPTQ_MARLIN_SUPPORTED_SYM = [True]


def get_scale_perms(num_bits: int):
    """column orders of `marlin_permute_scales` (gptq_marlin.py:26-35): inside a run of 64 columns for grouped scales,
    of 32 for channelwise ones -- the 8 columns an MMA lane needs end up contiguous"""
    grouped = [i + 8 * j for i in range(8) for j in range(8)]
    single = [2 * i + j for i in range(4) for j in (0, 1, 8, 9, 16, 17, 24, 25)]
    return grouped, single


def get_pack_factor(num_bits: int):
    assert num_bits in GPTQ_MARLIN_SUPPORTED_NUM_BITS, f"Unsupported num_bits = {num_bits}"
    return 32 // num_bits


def marlin_permute_scales(s: torch.Tensor, size_k: int, size_n: int, group_size: int, num_bits: int):
    """[groups, N] scales in natural column order -> the order the kernel reads (gptq_marlin.py:47-56)"""
    grouped, single = get_scale_perms(num_bits)
    order = grouped if (group_size != -1 and group_size < size_k) else single
    return s.reshape(-1, len(order))[:, order].reshape(-1, size_n).contiguous()


_SUPPORT = (   # (json key, attribute, admitted values) -- one table for the constructor and `is_marlin_compatible`
    ("bits", "weight_bits", GPTQ_MARLIN_SUPPORTED_NUM_BITS),
    ("group_size", "group_size", GPTQ_MARLIN_SUPPORTED_GROUP_SIZES),
    ("sym", "is_sym", GPTQ_MARLIN_SUPPORTED_SYM),
)


class GPTQMarlinConfig(QuantizationConfig):
    """`quantize_config.json`: {"bits": 4|8, "group_size": -1|32|64|128, "desc_act": bool, "sym": true, "lm_head": bool}"""

    def __init__(self, weight_bits: int, group_size: int, desc_act: bool, is_sym: bool, lm_head_quantized: bool) -> None:
        self.weight_bits, self.group_size, self.is_sym = weight_bits, group_size, is_sym
        self.desc_act = desc_act and group_size != -1      # one group per column: act-order is the identity
        self.lm_head_quantized = lm_head_quantized
        messages = {"weight_bits": "Marlin does not support weight_bits = {v}. Only weight_bits = {ok} are supported.",
                    "group_size": "Marlin does not support group_size = {v}. Only group_sizes = {ok} are supported.",
                    "is_sym": "Marlin does not support is_sym = {v}. Only sym = {ok} are supported."}
        for _, attr, admitted in _SUPPORT:
            if getattr(self, attr) not in admitted:
                raise ValueError(messages[attr].format(v=getattr(self, attr), ok=admitted))
        self.pack_factor = get_pack_factor(weight_bits)
        self.tile_size, self.max_parallel = GPTQ_MARLIN_TILE, GPTQ_MARLIN_MAX_PARALLEL
        self.min_thread_n, self.min_thread_k = GPTQ_MARLIN_MIN_THREAD_N, GPTQ_MARLIN_MIN_THREAD_K

    def __repr__(self) -> str:
        return (f"GPTQMarlinConfig(weight_bits={self.weight_bits}, group_size={self.group_size}, "
                f"desc_act={self.desc_act}, lm_head_quantized={self.lm_head_quantized})")

    @classmethod
    def get_name(cls) -> str:
        return "gptq_marlin"

    @classmethod
    def get_supported_act_dtypes(cls) -> List[torch.dtype]:
        return [torch.half, torch.bfloat16]

    @classmethod
    def get_min_capability(cls) -> int:
        return 80   # gfx950 reports 95

    @classmethod
    def get_config_filenames(cls) -> List[str]:
        return ["quantize_config.json"]

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "GPTQMarlinConfig":
        required = {k: cls.get_from_keys(config, [k]) for k in ("bits", "group_size", "desc_act", "sym")}
        return cls(required["bits"], required["group_size"], required["desc_act"], required["sym"],
                   cls.get_from_keys_or(config, ["lm_head"], default=False))

    @classmethod
    def is_marlin_compatible(cls, quant_config: Dict[str, Any]):
        """can a checkpoint written for plain GPTQ run here (gptq_marlin.py:160-184)"""
        if any(quant_config.get(key) is None for key in ("bits", "group_size", "sym", "desc_act")):
            return False
        major, minor = current_platform.get_device_capability()
        if major * 10 + minor < cls.get_min_capability():
            return False
        return all(quant_config[key] in admitted for key, _, admitted in _SUPPORT)

    @classmethod
    def override_quantization_method(cls, hf_quant_cfg, user_quant) -> Optional[str]:
        """a `gptq` checkpoint is claimed when it is compatible and the user did not insist on something else"""
        user_agrees = user_quant in (None, "marlin")
        return cls.get_name() if cls.is_marlin_compatible(hf_quant_cfg) and user_agrees else None

    def get_quant_method(self, layer: torch.nn.Module) -> Optional["GPTQMarlinLinearMethod"]:
        from ..linear import LinearBase
        from ..vocab_parallel_embedding import ParallelLMHead
        takes = isinstance(layer, LinearBase) or (self.lm_head_quantized and isinstance(layer, ParallelLMHead))
        return GPTQMarlinLinearMethod(self) if takes else None

    def get_scaled_act_names(self) -> List[str]:
        return []

    # ---- the schema -----------------------------------------------------------------------------------------
    def _group(self, g: Geometry) -> int:
        return g.k_all if self.group_size == -1 else self.group_size

    def _groups_sliced(self, g: Geometry) -> bool:
        """a row-parallel shard keeps only its own groups -- unless act-order, whose g_idx may name any group"""
        return g.row_sharded and self.group_size != -1 and not self.desc_act

    def is_k_full(self, g: Geometry) -> bool:
        return not (self.desc_act and g.row_sharded)

    def requirements(self) -> List[Require]:
        def divisible(what: str, value, unit_name: str, unit, spaces: str = " ") -> Require:
            return Require(lambda g: value(g) % unit(g) == 0,
                           lambda g: f"Weight {what} = {value(g)} is not divisible by{spaces}{unit_name} = {unit(g)}.")

        return [
            Require(lambda g: g.dtype in (torch.float16, torch.bfloat16),
                    lambda g: f"The params dtype must be float16 or bfloat16, but got {g.dtype}"),
            divisible("output_size_per_partition", lambda g: g.n, "min_thread_n", lambda g: self.min_thread_n, "  "),
            divisible("input_size_per_partition", lambda g: g.k, "min_thread_k", lambda g: self.min_thread_k),
            Require(lambda g: self._group(g) >= g.k_all or g.k % self._group(g) == 0,
                    lambda g: f"Weight input_size_per_partition = {g.k} is not divisible by group_size = {self._group(g)}."),
        ]

    def slots(self) -> List[Slot]:
        pack = self.pack_factor

        def groups(g: Geometry) -> int:
            return (g.k if self._groups_sliced(g) else g.k_all) // self._group(g)

        def group_axis(g: Geometry) -> Optional[int]:
            return 0 if self._groups_sliced(g) else None

        return [
            Slot("qweight", lambda g: (g.k // pack, g.n), torch.int32,
                 lambda g: {"input_dim": 0, "output_dim": 1, "packed_dim": 0, "pack_factor": pack}),
            Slot("g_idx", lambda g: (g.k, ), torch.int32, lambda g: {"input_dim": 0, "ignore_warning": True}),
            Slot("scales", lambda g: (groups(g), g.n), lambda g: g.dtype,
                 lambda g: {"input_dim": group_axis(g), "output_dim": 1}),
            Slot("qzeros", lambda g: (groups(g), g.n // pack), torch.int32,
                 lambda g: {"input_dim": group_axis(g), "output_dim": 1, "packed_dim": 1, "pack_factor": pack},
                 device="meta"),
        ]


class GPTQMarlinState(enum.Enum):
    REPACK = enum.auto()    # parameters still hold the GPTQ checkpoint layout
    READY = enum.auto()     # parameters hold the Marlin interchange layout


def _overwrite(param: torch.Tensor, new: torch.Tensor) -> None:
    """new contents and shape, same registered parameter (its storage is resized, not replaced)"""
    param.resize_(new.shape)
    param.copy_(new)


def _empty_index(device) -> Parameter:
    return Parameter(torch.empty(0, dtype=torch.int, device=device), requires_grad=False)


class GPTQMarlinLinearMethod(LinearMethodBase):

    NATIVE_MAX_M = 64   # rows per call up to which the MFMA-native copy is used (its M tiles end at 64 rows)

    def __init__(self, quant_config: GPTQMarlinConfig) -> None:
        self.quant_config = quant_config

    def create_weights(self, layer: torch.nn.Module, input_size_per_partition: int, output_partition_sizes: List[int],
                       input_size: int, output_size: int, params_dtype: torch.dtype, **extra_weight_attrs) -> None:
        cfg = self.quant_config
        g = Geometry(input_size_per_partition, tuple(output_partition_sizes), params_dtype, k_total=input_size)
        build(layer, g, cfg.requirements(), cfg.slots(), extra_weight_attrs)
        assert not cfg.desc_act or cfg.group_size != -1
        # plain attributes (not checkpoint tensors): the act-order permutation, the split-K tickets, the geometry
        layer.g_idx_sort_indices = torch.empty(g.k, dtype=torch.int32)
        layer.workspace = torch.zeros((g.n // cfg.min_thread_n) * cfg.max_parallel, dtype=torch.int)
        layer.input_size_per_partition, layer.output_size_per_partition, layer.input_size = g.k, g.n, g.k_all
        layer.is_k_full = cfg.is_k_full(g)
        layer.marlin_state = GPTQMarlinState.REPACK

    # ---- first call: GPTQ layout -> Marlin layout (gptq_marlin.py:389-447) ------------------------------------------
    def _repack(self, layer: torch.nn.Module) -> None:
        cfg = self.quant_config
        k, n = layer.input_size_per_partition, layer.output_size_per_partition
        dev = layer.qweight.device
        layer.workspace = layer.workspace.to(dev)
        if cfg.desc_act:    # rows sorted by group: the kernel walks whole groups, `perm` gathers the activations
            order = torch.argsort(layer.g_idx).to(torch.int)
            layer.g_idx_sort_indices = layer.g_idx_sort_indices.to(dev)
            _overwrite(layer.g_idx, layer.g_idx[order])
            _overwrite(layer.g_idx_sort_indices, order)
        else:
            layer.g_idx, layer.g_idx_sort_indices = _empty_index(dev), _empty_index(dev)
        if self.native_eligible(layer):   # built from the same GPTQ words
            # buffers, not plain attributes: module.to() / memory accounting see them; persistent=False keeps them out of
            # the state dict (they are derived data)
            for name, t in (("qweight_native", ops.w4_native_repack(layer.qweight.data, None, k, n)),
                            ("scales_native", layer.scales.data.clone())):     # natural [groups, N]
                if name in layer._buffers:
                    layer._buffers[name] = t
                else:
                    layer.register_buffer(name, t, persistent=False)
            if os.environ.get("NMV_W4_KEEP_MARLIN", "0") != "1":
                # ONE weight tensor: nmv_w4_native_gemm serves every row count (decode: stream / ring kernels; prompt-sized:
                # csrc/w4a16_prefill.hip), so the Marlin tensor and its permuted scales are not built and the GPTQ words
                # are released -- 0.5 byte per weight on the rank instead of 1.  (NMV_W4_KEEP_MARLIN=1 keeps both and
                # sends each call where it was measured faster: 3-6 % on 65 .. 256-row calls.)
                layer.qweight.data = torch.empty(0, dtype=layer.qweight.dtype, device=dev)
                layer.scales.data = torch.empty(0, dtype=layer.scales.dtype, device=dev)
                layer.marlin_dropped = True
                return
            extra = layer.qweight_native.numel() * 4 + layer.scales_native.numel() * layer.scales_native.element_size()
            type(self).native_extra_bytes = getattr(type(self), "native_extra_bytes", 0) + extra
        _overwrite(layer.qweight, ops.gptq_marlin_repack(layer.qweight, layer.g_idx_sort_indices, k, n, cfg.weight_bits))
        scales_k = layer.input_size if cfg.desc_act else k
        _overwrite(layer.scales, marlin_permute_scales(layer.scales, scales_k, n, cfg.group_size, cfg.weight_bits))

    def _ready(self, layer: torch.nn.Module, interleave_gate_up: bool = False) -> None:
        if layer.marlin_state is not GPTQMarlinState.REPACK:
            return
        if interleave_gate_up:
            assert self.can_fuse_silu_mul(layer)
        layer.marlin_state = GPTQMarlinState.READY
        if interleave_gate_up:
            layer.qweight.data = self._interleave_gate_up(layer.qweight.data)
            layer.scales.data = self._interleave_gate_up(layer.scales.data)
            layer.gate_up_interleaved = True
        self._repack(layer)

    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        return   # the conversion is lazy (first call), as in the reference: the loader may still be writing shards

    # ---- the reference's forward ---------------------------------------------------------------------------------
    def apply(self, layer: torch.nn.Module, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        assert not getattr(layer, "gate_up_interleaved", False), "this layer's columns are interleaved for apply_silu_mul()"
        self._ready(layer)
        rows = x.reshape(-1, x.shape[-1])
        m, n, k = rows.shape[0], layer.output_size_per_partition, layer.input_size_per_partition
        if self._native(layer, m):
            y = ops.w4_native_gemm(rows, layer.qweight_native, layer.scales_native, layer.workspace, m, n, k, mode=0)
        else:
            y = ops.gptq_marlin_gemm(rows, layer.qweight, layer.scales, layer.g_idx, layer.g_idx_sort_indices,
                                     layer.workspace, self.quant_config.weight_bits, m, n, k, layer.is_k_full)
        if bias is not None:
            y.add_(bias)
        return y.reshape(x.shape[:-1] + (n, ))

    # ---- MFMA-native copy of the codes (optional) -------------------------------------------------------------------
    def _plain_w4(self, layer: torch.nn.Module) -> bool:
        """4-bit, group 128 or channelwise, no act-order, whole K on this rank in 256-k rings: what the fused forms
        and the native copy are written for"""
        cfg = self.quant_config
        return (cfg.weight_bits == 4 and not cfg.desc_act and cfg.group_size in (-1, 128)
                and layer.input_size_per_partition % 256 == 0 and layer.is_k_full)

    def native_eligible(self, layer: torch.nn.Module) -> bool:
        return (os.environ.get("NMV_W4_NATIVE", "1") != "0" and self._plain_w4(layer) and self.quant_config.group_size == 128
                and layer.output_size_per_partition % 64 == 0 and layer.qweight.is_cuda)

    @classmethod
    def _native(cls, layer, size_m: int) -> bool:
        """measured on MI355X (tools/bench_gemm.py --native, tools/sweep_ring.py): every decode-sized call takes the native
        tensor -- up to 16 rows csrc/w4a16_stream.hip's resident form (3-7 % over the Marlin form), 17..32 rows the ring
        kernel (csrc/w4a16_ring.hip: gate_up 18.5 us against 25.8 on the Marlin tensor at M = 32), 33..64 rows the ring on the
        wide projection and the stream kernel's native form elsewhere.  Prompt-sized calls: csrc/w4a16_prefill.hip (256 x 256
        or 128 x 128 tiles; gate_up M = 512: 128 us against 165-173 on the Marlin tensor, level with it on the narrow
        projections).  By default the layer keeps ONLY the native tensor and every call comes here; with
        NMV_W4_KEEP_MARLIN=1 it keeps both and a prompt-sized call goes where it was measured faster
        (nmv_w4_native_prefill_plan)."""
        if getattr(layer, "qweight_native", None) is None:
            return False
        if size_m <= cls.NATIVE_MAX_M or getattr(layer, "marlin_dropped", False):
            return True
        return ops.w4_native_prefill_plan(size_m, layer.output_size_per_partition, layer.input_size_per_partition)

    # ---- deferred split-K: the GEMM leaves fp32 slabs, the following launch sums them ------------------------------
    def can_defer_reduce(self, layer: torch.nn.Module) -> bool:
        return self._plain_w4(layer) and not getattr(layer, "gate_up_interleaved", False)

    def can_defer(self, layer: torch.nn.Module, rows: int) -> bool:
        """can_defer_reduce() and the kernel has a deferred plan for `rows` rows of THIS layer: asked of the planner that
        apply_partial will run (native or Marlin tensor) with the layer's own group count (channelwise layers have one)"""
        if not self.can_defer_reduce(layer):
            return False
        if layer.qweight.is_cuda:
            self._ready(layer)   # the first-call repack decides which tensor apply_partial will use
        n, k = layer.output_size_per_partition, layer.input_size_per_partition
        groups = int(layer.scales.shape[0]) if layer.scales.dim() == 2 else max(k // 128, 1)
        if self._native(layer, rows):
            return ops.w4_native_gemm_splits(rows, n, k, groups) >= 1
        return ops.gptq_marlin_gemm_partial_splits(rows, n, k, groups) >= 1

    def apply_partial(self, layer: torch.nn.Module, x: torch.Tensor, allow16: bool = False) -> torch.Tensor:
        """x @ W as fp32 split-K slabs [splits, T, N]; their sum in split order, rounded to the model dtype, is
        bit-identical to apply().  allow16 (the caller's consumer reads slabs of either width: the norm and the rope +
        cache launches): a prompt-sized call leaves its slabs in the MODEL dtype -- at 512 rows the slabs are a fifth of
        the layer's bytes, and half of that is saved; each slab is rounded once before the fp32 sum (within the op's
        tolerance, not bit-identical to apply())"""
        self._ready(layer)
        rows = x.reshape(-1, x.shape[-1])
        m, n, k = rows.shape[0], layer.output_size_per_partition, layer.input_size_per_partition
        if self._native(layer, m):
            mode = 3 if (allow16 and m > self.NATIVE_MAX_M and os.environ.get("NMV_W4_SLAB16", "1") != "0"
                         and ops.w4_native_gemm_slab16(m, n, k)) else 2
            return ops.w4_native_gemm(rows, layer.qweight_native, layer.scales_native, None, m, n, k, mode=mode)
        return ops.gptq_marlin_gemm_partial(rows, layer.qweight, layer.scales, m, n, k)

    # ---- gate_up with silu_and_mul in the GEMM's epilogue -----------------------------------------------------------
    def can_fuse_silu_mul(self, layer: torch.nn.Module) -> bool:
        """a merged [gate | up] projection whose columns can still be interleaved per 64-column chunk (i.e. before the
        first-call repack), wide enough that a GEMM without split-K -- the epilogue needs the whole K in one workgroup
        -- beats GEMM + silu_and_mul: from 224 chunks (N = 14336: Llama-3-8B at TP <= 2) always, from 112 while the
        whole K fits a workgroup's LDS"""
        if getattr(layer, "gate_up_interleaved", False):
            return True
        n = layer.output_size_per_partition
        # round 3, re-measured on the stream kernel (tools/bench_gemm.py --native --shapes gate_up_tp2,gate_up_tp4,gate_up_tp8,
        # gate_up70): the fused form never splits K, so below 224 chunks it fills a fraction of the CUs (112 chunks = 56
        # workgroups) while the plain GEMM splits K to 256 -- at N = 7168 (Llama-3-8B at TP = 4) fused 15.5 / 16.2 / 19.0 us
        # at M = 1 / 16 / 64 against 9.2 / 10.0 / 16.7 + a 4.5 us silu_and_mul launch, and at K = 8192 (Llama-3-70B at
        # TP = 8) it leaves the stream kernel altogether (28.2 us against 12.7).  From 224 chunks on (activations
        # streamed, one k range per workgroup) it costs 0.3-1.2 us over the plain GEMM and saves the launch.
        # Between 112 and 223 chunks it stays on where the whole K fits a workgroup's LDS (K <= 4096), because at the
        # headline batch it still wins (M = 64: 16.5 us on the Marlin tensor against 21.2) -- apply_silu_mul picks the
        # tensor per call there.
        k = layer.input_size_per_partition
        wide_enough = n // 64 >= 224 or (n // 64 >= 112 and k <= 4096)
        return (layer.marlin_state is GPTQMarlinState.REPACK and self._plain_w4(layer) and n % 128 == 0
                and wide_enough and getattr(layer, "bias", None) is None)

    @staticmethod
    def _interleave_gate_up(t: torch.Tensor) -> torch.Tensor:
        """columns [gate 0..I-1 | up 0..I-1] -> per 64-column chunk c: [gate 32c..32c+31 | up 32c..32c+31]"""
        lead, n = t.shape[:-1], t.shape[-1]
        return t.reshape(*lead, 2, n // 64, 32).transpose(-3, -2).reshape(*lead, n).contiguous()

    def apply_silu_mul(self, layer: torch.nn.Module, x: torch.Tensor) -> torch.Tensor:
        """silu(x @ W_gate) * (x @ W_up) -> [.., N / 2]: the roundings of apply() + SiluAndMul (gate and up rounded to the
        model dtype, silu rounded, product rounded).  Bit-identical to that sequence when the plain GEMM does not split K
        across workgroups; where it does (112 .. 223 chunks) the two differ in the order of the fp32 sums only -- the fused
        launch never splits -- i.e. by at most an ulp of the model dtype on a few elements (tests/test_gpu_w4_native.py:
        test_native_silu_mul_on_a_split_shape)"""
        self._ready(layer, interleave_gate_up=True)
        assert getattr(layer, "gate_up_interleaved", False), "layer was repacked without the interleave"
        rows = x.reshape(-1, x.shape[-1])
        m, n, k = rows.shape[0], layer.output_size_per_partition, layer.input_size_per_partition
        # narrow fused launches (112 .. 223 chunks, resident form without split-K): the native tensor only wins up to a
        # few rows (M = 1: 15.5 vs 16.8 us; M = 16: 16.2 vs 12.2; M = 64: 19.0 vs 16.5)
        if self._native(layer, m) and (m > self.NATIVE_MAX_M or n // 64 >= 224 or m <= 8):
            y = ops.w4_native_gemm(rows, layer.qweight_native, layer.scales_native, layer.workspace, m, n, k, mode=1)
        else:
            y = ops.gptq_marlin_gemm_silu_mul(rows, layer.qweight, layer.scales, layer.workspace, m, n, k)
        return y.reshape(x.shape[:-1] + (n // 2, ))
