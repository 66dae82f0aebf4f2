"""Decode-loop harness: the minimum of the reference's worker / model-runner / cache-engine that is
needed to drive the hot path end to end with synthetic weights.

What it reproduces (and nothing more):
  * KV-cache allocation per layer, shape [2, num_blocks, block_size*kv_heads*head_size]
    (vllm/worker/cache_engine.py:70-88, PagedAttention.get_kv_cache_shape);
  * the per-step inputs of ModelRunner._prepare_model_input_tensors (vllm/worker/model_runner.py:
    332-640): input_ids, positions, slot_mapping (slot = block_table[pos // bs] * bs + pos % bs,
    :572-580), seq_lens, block_tables and the attention metadata;
  * greedy sampling of the next token (argmax of the logits);
  * hipGraph capture of one decode step (the reference captures decode batches in
    CUDAGraphRunner, model_runner.py:910-1118; torch.cuda.CUDAGraph is hipGraph on ROCm).
    The captured step also advances positions / seq_lens / slot_mapping ON DEVICE, so a whole
    decode loop replays without host work.
There is no scheduler, block manager, tokenizer or checkpoint loader: those sit above the
drop-in boundary and stay the reference's own.
"""
import os
import sys
import time
from dataclasses import dataclass
from typing import List, Optional

import torch

from .. import _custom_ops as ops
from ..attention.backends.rocm_hip_attn import ROCmHipAttentionMetadata
from ..distributed import (get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size,
                           get_tp_group, tensor_model_parallel_all_gather)
from ..model_executor.layers.quantization import get_quantization_config
from ..model_executor.models.llama import LlamaForCausalLM


@dataclass
class LlamaArch:
    hidden_size: int
    intermediate_size: int
    num_hidden_layers: int
    num_attention_heads: int
    num_key_value_heads: int
    vocab_size: int
    rms_norm_eps: float = 1e-5
    rope_theta: float = 500000.0
    max_position_embeddings: int = 8192
    hidden_act: str = "silu"
    rope_scaling: Optional[dict] = None
    tie_word_embeddings: bool = False

    @property
    def head_dim(self):
        return self.hidden_size // self.num_attention_heads


LLAMA3_8B = LlamaArch(4096, 14336, 32, 32, 8, 128256)
LLAMA3_70B = LlamaArch(8192, 28672, 80, 64, 8, 128256)
TINY = LlamaArch(512, 1024, 2, 8, 2, 2048)
# the per-rank geometry of Llama-3-70B at TP = 8 (8 query heads + 1 KV head per rank, BASELINE.json configs[4])
# at TP = 2 and toy widths: what the one-GPU rehearsal of `bench.py --model llama3-70b --gpus 8` runs
TINY_70B = LlamaArch(1024, 3584, 2, 16, 2, 2048)


class CaptureFailedError(RuntimeError):
    """hipGraph capture of the decode step failed on at least one rank of a tensor-parallel group.  The process
    must not go on issuing collectives: round 1's SIGSEGV was exactly "capture fails, the process keeps going
    eagerly".  The caller exits non-zero; a parent that never touched the GPU may start fresh ranks with
    --no-graph (bench.py does)."""


def agree_on_capture(ok: bool, world: int, cpu_group) -> bool:
    """every rank replays a graph, or none does: gathers the ranks' capture verdicts over the CPU group (no device
    collective right after a capture).  Raises CaptureFailedError on EVERY rank when any rank's capture failed."""
    if world <= 1:
        return ok
    import torch.distributed as dist
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok), group=cpu_group)
    if not all(flags):
        bad = [r for r, f in enumerate(flags) if not f]
        raise CaptureFailedError(f"hipGraph capture of the decode step failed on rank(s) {bad}; a tensor-parallel "
                                 "group does not continue eagerly in the same processes -- rerun with --no-graph")
    return True


@dataclass
class CacheConfig:
    block_size: int = 16
    cache_dtype: str = "auto"
    sliding_window: Optional[int] = None


def gptq_quantize_on_device(w: torch.Tensor, bits: int, group_size: int):
    """Symmetric group quantisation + GPTQ packing with torch ops on the weight's device
    (the semantics of the reference's quantize_weights / gptq_pack, quant_utils.py:39-146).
    Host-side setup of synthetic checkpoints: plumbing, not the measured path."""
    k, n = w.shape
    gs = k if group_size == -1 else group_size
    maxq = 2**bits - 1
    half = (maxq + 1) // 2
    wg = w.float().reshape(k // gs, gs, n)
    s = wg.abs().amax(dim=1, keepdim=True).clamp_min(1e-8) * (2.0 / maxq)
    q = torch.clamp(torch.round(wg / s) + half, 0, maxq).to(torch.int64).reshape(k, n)
    pf = 32 // bits
    shifts = (torch.arange(pf, device=w.device, dtype=torch.int64) * bits).view(1, pf, 1)
    packed = (q.view(k // pf, pf, n) << shifts).sum(dim=1)
    packed = torch.where(packed >= 2**31, packed - 2**32, packed).to(torch.int32)
    g_idx = (torch.arange(k, device=w.device, dtype=torch.int32) // gs)
    return packed, s.reshape(k // gs, n).to(w.dtype), g_idx


def synthetic_llama_weights(arch: LlamaArch, dtype: torch.dtype, device, quant: Optional[dict],
                            seed: int = 0):
    """Yields (checkpoint-style name, tensor) for a random-init Llama: W ~ N(0, 0.02^2)
    (BASELINE.md section 3), quantised to GPTQ int4 when `quant` is given.  Every TP rank
    generates the same full tensors (same seeds) and keeps its shard through the weight loaders."""
    gen = torch.Generator(device=device)
    h, inter, hd = arch.hidden_size, arch.intermediate_size, arch.head_dim
    nq, nkv = arch.num_attention_heads, arch.num_key_value_heads

    def dense(name, out_f, in_f, sd):
        gen.manual_seed(seed * 100003 + sd)
        w = torch.randn((out_f, in_f), generator=gen, device=device, dtype=torch.float32) * 0.02
        return name, w.to(dtype)

    def linear(prefix, out_f, in_f, sd):
        gen.manual_seed(seed * 100003 + sd)
        # GPTQ stores W^T: [in_features, out_features]
        w = (torch.randn((in_f, out_f), generator=gen, device=device, dtype=torch.float32) * 0.02)
        if quant is None:
            yield prefix + ".weight", w.t().contiguous().to(dtype)
            return
        if quant.get("method") == "w8a8":
            # compressed-tensors W8A8: int8 [out, in], per-output-channel fp32 scales, symmetric
            wt = w.t().contiguous()
            ws = wt.abs().amax(dim=1, keepdim=True).clamp_min(1e-8) / 127.0
            yield prefix + ".weight", torch.clamp(torch.round(wt / ws), -127, 127).to(torch.int8)
            yield prefix + ".weight_scale", ws.to(torch.float32)
            return
        qw, s, g_idx = gptq_quantize_on_device(w.to(dtype), quant["bits"], quant["group_size"])
        yield prefix + ".qweight", qw
        yield prefix + ".scales", s
        yield prefix + ".g_idx", g_idx

    yield dense("model.embed_tokens.weight", arch.vocab_size, h, 1)
    for i in range(arch.num_hidden_layers):
        p = f"model.layers.{i}."
        yield from linear(p + "self_attn.q_proj", nq * hd, h, 10 + 16 * i)
        yield from linear(p + "self_attn.k_proj", nkv * hd, h, 11 + 16 * i)
        yield from linear(p + "self_attn.v_proj", nkv * hd, h, 12 + 16 * i)
        yield from linear(p + "self_attn.o_proj", h, nq * hd, 13 + 16 * i)
        yield from linear(p + "mlp.gate_proj", inter, h, 14 + 16 * i)
        yield from linear(p + "mlp.up_proj", inter, h, 15 + 16 * i)
        yield from linear(p + "mlp.down_proj", h, inter, 16 + 16 * i)
        yield p + "input_layernorm.weight", torch.ones(h, dtype=dtype, device=device)
        yield p + "post_attention_layernorm.weight", torch.ones(h, dtype=dtype, device=device)
    yield "model.norm.weight", torch.ones(h, dtype=dtype, device=device)
    yield dense("lm_head.weight", arch.vocab_size, h, 2)


class DecodeRunner:
    """One model replica (or TP shard) + its KV cache + a captured decode step."""

    @classmethod
    def from_pretrained(cls, model_dir: str, device: torch.device, dtype: Optional[torch.dtype] = None,
                        cache_config: Optional[CacheConfig] = None) -> "DecodeRunner":
        """a runner on a local HF-layout Llama checkpoint: config.json -> architecture, the checkpoint's
        quantisation config -> the reference's method (GPTQ -> gptq_marlin when the Marlin kernels can run
        it), *.safetensors shards -> LlamaForCausalLM.load_weights (model_executor/model_loader.py)"""
        from ..model_executor import model_loader as ml
        cfg = ml.read_hf_config(model_dir)
        arch = LlamaArch(cfg["hidden_size"], cfg["intermediate_size"], cfg["num_hidden_layers"],
                         cfg["num_attention_heads"], cfg.get("num_key_value_heads", cfg["num_attention_heads"]),
                         cfg["vocab_size"], cfg.get("rms_norm_eps", 1e-5), cfg.get("rope_theta", 10000.0),
                         cfg.get("max_position_embeddings", 8192), cfg.get("hidden_act", "silu"),
                         cfg.get("rope_scaling"), bool(cfg.get("tie_word_embeddings", False)))
        if dtype is None:
            # config.py:_get_and_verify_dtype: fp32 checkpoints are served in fp16, fp16 / bf16 as they are
            td = str(cfg.get("torch_dtype", "float16")).replace("torch.", "")
            known = {"bfloat16": torch.bfloat16, "float16": torch.float16, "float32": torch.float16}
            if td not in known:
                raise ValueError(f"config.json: unsupported torch_dtype {td!r} (float16 / bfloat16 / float32)")
            dtype = known[td]
        return cls(arch, device, dtype, None, cache_config, weights=ml.safetensors_weights_iterator(model_dir),
                   quant_config=ml.build_quant_config(model_dir, cfg))

    def __init__(self, arch: LlamaArch, device: torch.device, dtype: torch.dtype = torch.bfloat16,
                 quant: Optional[dict] = None, cache_config: Optional[CacheConfig] = None,
                 seed: int = 0, weights=None, quant_config=None):
        self.arch = arch
        self.device = device
        self.dtype = dtype
        self.cache_config = cache_config or CacheConfig()
        self.tp_size = get_tensor_model_parallel_world_size()
        self.fused_step_tail = os.environ.get("NMV_FUSED_GLUE", "1") != "0"
        self.keep_logits = False
        self.last_logits: Optional[torch.Tensor] = None
        self.tp_rank = get_tensor_model_parallel_rank()
        if quant_config is not None:
            pass  # the checkpoint's own QuantizationConfig (from_pretrained)
        elif quant is not None and quant.get("method") == "w8a8":
            # BASELINE.json configs[3]: int8 weights (per channel) x int8 activations (dynamic per token)
            quant_config = get_quantization_config("compressed-tensors").from_config({
                "format": "int-quantized",
                "config_groups": {"group_0": {
                    "targets": ["Linear"],
                    "weights": dict(num_bits=8, type="int", strategy="channel", symmetric=True, dynamic=False),
                    "input_activations": dict(num_bits=8, type="int", strategy="token", symmetric=True,
                                              dynamic=True)}},
                "ignore": ["lm_head"]})
        elif quant is not None:
            quant_config = get_quantization_config(quant.get("method", "gptq_marlin")).from_config(
                dict(bits=quant["bits"], group_size=quant["group_size"],
                     desc_act=quant.get("desc_act", False), sym=True))
        prev = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        try:
            with torch.device(device):
                self.model = LlamaForCausalLM(arch, self.cache_config, quant_config)
        finally:
            torch.set_default_dtype(prev)
        if weights is None:
            weights = synthetic_llama_weights(arch, dtype, device, quant, seed)
        self.model.load_weights(weights)
        for m in self.model.modules():
            qm = getattr(m, "quant_method", None)
            if qm is not None:
                qm.process_weights_after_loading(m)
        self.num_kv_heads = max(1, arch.num_key_value_heads // self.tp_size)
        self.num_heads = arch.num_attention_heads // self.tp_size
        self.kv_caches: List[torch.Tensor] = []
        self.graph = None

    # ------------------------------------------------------------------ KV cache
    def allocate_kv_cache(self, num_blocks: int) -> None:
        bs = self.cache_config.block_size
        cdt = torch.uint8 if self.cache_config.cache_dtype != "auto" else self.dtype
        shape = (2, num_blocks, bs * self.num_kv_heads * self.arch.head_dim)
        self.kv_caches = [torch.zeros(shape, dtype=cdt, device=self.device)
                          for _ in range(self.arch.num_hidden_layers)]
        self.num_blocks = num_blocks

    # ------------------------------------------------------------------ batch state
    def setup_batch(self, batch: int, context_len: int, max_new_tokens: int, seed: int = 0) -> None:
        """`batch` sequences that each already hold `context_len` tokens in the cache; block tables
        are a random permutation of the block ids."""
        bs = self.cache_config.block_size
        # positions index the rotary cos/sin table: past max_position_embeddings the kernels would read
        # beyond it (the reference's scheduler never lets a sequence grow that far, config.py max_model_len)
        if context_len + max_new_tokens > self.arch.max_position_embeddings:
            raise ValueError(f"context {context_len} + {max_new_tokens} new tokens exceeds the model's "
                             f"max_position_embeddings = {self.arch.max_position_embeddings}")
        self.batch = batch
        self.max_seq_len = context_len + max_new_tokens
        self._steps_left = max_new_tokens
        blocks_per_seq = (self.max_seq_len + bs - 1) // bs
        need = batch * blocks_per_seq
        if not self.kv_caches or self.num_blocks < need:
            self.allocate_kv_cache(need + need // 4 + 1)
        g = torch.Generator().manual_seed(seed)
        perm = torch.randperm(self.num_blocks, generator=g)[:need].to(torch.int32)
        self.block_tables = perm.view(batch, blocks_per_seq).to(self.device)
        self.input_ids = torch.randint(0, self.arch.vocab_size, (batch, ), generator=g).to(self.device)
        # the token being decoded sits at position context_len (0-based); its K/V is written first
        self.positions = torch.full((batch, ), context_len, dtype=torch.int64, device=self.device)
        self.seq_lens = torch.full((batch, ), context_len + 1, dtype=torch.int32, device=self.device)
        self.slot_mapping = torch.empty((batch, ), dtype=torch.int64, device=self.device)
        self._update_slots()
        self.graph = None

    def _update_slots(self) -> None:
        bs = self.cache_config.block_size
        blk = torch.gather(self.block_tables, 1, (self.positions // bs).view(-1, 1)).view(-1)
        self.slot_mapping.copy_(blk.to(torch.int64) * bs + self.positions % bs)

    def fill_context(self, seed: int = 1) -> None:
        """random K/V for the context tokens (instead of running a prefill)"""
        g = torch.Generator(device=self.device).manual_seed(seed)
        scale = self.arch.head_dim**-0.5
        for kv in self.kv_caches:
            if kv.dtype == torch.uint8:
                kv.copy_(torch.randint(0, 120, kv.shape, generator=g, device=self.device,
                                       dtype=torch.uint8))
            else:
                kv.copy_(((torch.rand(kv.shape, generator=g, device=self.device) * 2 - 1) * scale
                          ).to(kv.dtype))

    def _decode_metadata(self) -> ROCmHipAttentionMetadata:
        return ROCmHipAttentionMetadata(
            num_prefills=0, num_prefill_tokens=0, num_decode_tokens=self.batch,
            slot_mapping=self.slot_mapping, seq_lens=None, seq_lens_tensor=self.seq_lens,
            max_query_len=None, max_prefill_seq_len=0, max_decode_seq_len=self.max_seq_len,
            query_start_loc=None, seq_start_loc=None, context_lens_tensor=None,
            block_tables=self.block_tables, use_cuda_graph=self.graph is not None)

    # ------------------------------------------------------------------ steps
    def _sample(self, hidden_states: torch.Tensor) -> torch.Tensor:
        """greedy: argmax over the full vocabulary (logits of all TP shards gathered)"""
        logits = torch.matmul(hidden_states, self.model.lm_head.weight.t())
        if self.tp_size > 1:
            logits = tensor_model_parallel_all_gather(logits)
        if self.keep_logits:
            self.last_logits = logits[:, :self.arch.vocab_size].float().clone()
        return torch.argmax(logits[:, :self.arch.vocab_size], dim=-1)

    def _step_body(self) -> torch.Tensor:
        hidden = self.model(self.input_ids, self.positions, self.kv_caches, self._decode_metadata())
        if self.fused_step_tail:
            # argmax + state advance in two launches (csrc/sampling.hip) instead of torch.argmax and
            # five element-wise kernels
            logits = torch.matmul(hidden, self.model.lm_head.weight.t())
            car = get_tp_group().custom_ar if self.tp_size > 1 else None
            if car is not None:
                # vocab-parallel lm_head without gathering the logits: per-shard argmax record ->
                # P2P all-gather of B x 8 bytes per rank -> winner + state advance (no RCCL in the step)
                shard = logits.shape[1]
                lo = get_tensor_model_parallel_rank() * shard
                valid = max(0, min(shard, self.arch.vocab_size - lo))
                if valid == 0:  # a shard that is all padding never wins
                    rec = ops.greedy_sample_shard(torch.full_like(logits[:, :1], float("-inf")), lo)
                else:
                    rec = ops.greedy_sample_shard(logits[:, :valid], lo)
                return ops.greedy_sample_finish(car.all_gather_record(rec), self.tp_size, logits.shape[0],
                                                self.input_ids, self.positions, self.seq_lens,
                                                self.slot_mapping, self.block_tables,
                                                self.cache_config.block_size)
            if self.tp_size > 1:
                logits = tensor_model_parallel_all_gather(logits)
            if self.keep_logits:  # tests: the logits the token was drawn from (eager steps only)
                self.last_logits = logits[:, :self.arch.vocab_size].float().clone()
            return ops.greedy_sample_advance(logits[:, :self.arch.vocab_size], self.input_ids, self.positions,
                                             self.seq_lens, self.slot_mapping, self.block_tables,
                                             self.cache_config.block_size)
        next_tokens = self._sample(hidden)
        # advance the batch state on device so that a captured step can be replayed back to back
        self.input_ids.copy_(next_tokens)
        self.positions.add_(1)
        self.seq_lens.add_(1)
        self._update_slots()
        return next_tokens

    @torch.inference_mode()
    def decode_step(self) -> torch.Tensor:
        # the batch was set up with room for a fixed number of new tokens (block tables, rotary table):
        # one step more would index past them on the device
        self._steps_left -= 1
        if self._steps_left < 0:
            raise RuntimeError("decode_step: the batch has used up the max_new_tokens it was set up with")
        if self.graph is not None:
            self.graph.replay()
            return self._graph_out
        return self._step_body()

    def step_is_capturable(self) -> bool:
        """whether one decode step can be recorded into a hipGraph.  Without TP always; under TP when the
        step holds no process-group collective at all (row-parallel all-reduces and the sampler's gather
        on the P2P communicator) or when the device group is RCCL, whose collectives are captured by
        torch like kernels.  Any other backend (gloo on GPU tensors: the one-GPU rehearsal) is NOT: its
        collectives hand the tensor to a helper thread that synchronises a side stream, and a stream
        synchronisation inside a capture invalidates it -- after which the runtime's stream state is
        undefined (round 1: SIGSEGV in the first eager all_reduce that followed).  Such a step is never
        offered to torch.cuda.graph; it runs eagerly."""
        if self.tp_size == 1:
            return True
        tp = get_tp_group()
        if tp.backend == "nccl":
            return True
        # every collective of THIS step must take the P2P path -- a message it declines (larger than its buffers,
        # not a multiple of 16 bytes, or the communicator disabled after an error) would fall back to the process
        # group inside the capture: the [batch, hidden] activations of the row-parallel layers and the sampler's
        # (value, index) records
        car = tp.custom_ar
        if car is None or not self.fused_step_tail or not car.enabled:
            return False
        act_bytes = getattr(self, "batch", 1) * self.arch.hidden_size * 2
        rec_bytes = getattr(self, "batch", 1) * 8
        return 0 < act_bytes <= car.max_bytes and act_bytes % 16 == 0 and rec_bytes <= car.max_bytes

    @torch.inference_mode()
    def capture(self, warmup: int = 2) -> bool:
        """capture one decode step into a hipGraph; returns False (and stays eager) when the step is not
        capturable (step_is_capturable) or the capture fails"""
        self.graph = None
        if not self.step_is_capturable():
            if self.tp_rank == 0:
                print(f"[decode_runner] decode step not captured: it contains '{get_tp_group().backend}' "
                      "collectives, which cannot be recorded into a hipGraph; running eagerly", file=sys.stderr)
            return False
        # the capture runs warmup + 1 steps before the state is restored: they must fit into what the
        # batch was set up with (block tables, rotary table)
        if self._steps_left < warmup + 1:
            warmup = max(0, self._steps_left - 1)
            if self._steps_left < 1:
                raise RuntimeError("capture: the batch has no decode step left (max_new_tokens used up)")
        saved = [t.clone() for t in (self.input_ids, self.positions, self.seq_lens, self.slot_mapping)]
        prev_stream = torch.cuda.current_stream(self.device)
        if self.tp_size > 1:
            # the ranks enter the warm-up steps together: the P2P collectives wait for their peers
            # with a bounded spin, and ranks may have taken different times to build their weights
            torch.cuda.synchronize(self.device)
            get_tp_group().barrier()
        try:
            # tests: inject a failure on one rank BEFORE the eager warm-up -- the hard case: its peers enter the warm-up
            # without it.  Their first P2P collective runs into its bounded wait once; every later rendezvous of the
            # communicator returns at once (csrc/custom_all_reduce.hip: ar_rendezvous fails fast on the sticky error word),
            # and the error word is read after the FIRST eager step, so a lost peer costs about one bound, not one per
            # collective of every warm-up step (~65 each)
            if os.environ.get("NMV_TEST_FAIL_CAPTURE_RANK") == str(self.tp_rank):
                raise RuntimeError("injected capture failure (NMV_TEST_FAIL_CAPTURE_RANK)")
            s = torch.cuda.Stream(device=self.device)
            s.wait_stream(prev_stream)
            with torch.cuda.stream(s):
                for w in range(warmup):  # first touch: marlin repack, workspace moves, allocator
                    self._step_body()
                    if w == 0 and self.tp_size > 1:
                        torch.cuda.synchronize(self.device)
                        car = getattr(get_tp_group(), "custom_ar", None)
                        if car is not None and car.enabled and car.local_error():
                            raise RuntimeError("a P2P all-reduce of the warm-up timed out (a peer is missing)")
            prev_stream.wait_stream(s)
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            # thread_local: helper threads of the process group (watchdog, event polling) may call
            # into the runtime while this thread captures
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                out = self._step_body()
            self.graph, self._graph_out = graph, out
            ok = True
        except Exception as e:  # pragma: no cover - depends on the runtime
            # stderr: bench.py's stdout carries exactly one JSON line
            print(f"[decode_runner] hipGraph capture failed, staying eager: {str(e).splitlines()[0]}",
                  file=sys.stderr)
            self.graph = None
            ok = False
            # torch.cuda.graph.__exit__ raises from capture_end() before it restores the stream, so
            # the poisoned capture stream would stay current: go back to the caller's stream, then
            # drain the error code the invalidated capture left behind for the next checked HIP call
            torch.cuda.set_stream(prev_stream)
            for _ in range(8):
                try:
                    torch.cuda.synchronize(self.device)
                    break
                except Exception:
                    pass
        for dst, src in zip((self.input_ids, self.positions, self.seq_lens, self.slot_mapping), saved):
            dst.copy_(src)
        torch.cuda.synchronize(self.device)
        if self.tp_size > 1:
            # every rank replays a graph, or none does, and a failed capture is fatal for the whole group (agreed
            # over the CPU group: no device collective right after a capture); so is a P2P flag wait that ran out
            # during the warm-up
            try:
                agree_on_capture(ok, self.tp_size, get_tp_group().cpu_group)
            except CaptureFailedError:
                self.graph = None
                raise
            get_tp_group().check_custom_ar_error()
        return ok

    def check_collectives(self) -> None:
        """raise when a P2P all-reduce of this runner's group timed out since the last check (its
        outputs, and every token since, are NaN / garbage); collective over the TP group"""
        if self.tp_size > 1:
            get_tp_group().check_custom_ar_error()

    @torch.inference_mode()
    def prefill(self, prompt_len: int, seed: int = 0) -> torch.Tensor:
        """one prompt step for `batch` prompts of `prompt_len` tokens (TTFT); K/V go to the cache"""
        bs = self.cache_config.block_size
        b = self.batch
        g = torch.Generator().manual_seed(seed)
        ids = torch.randint(0, self.arch.vocab_size, (b * prompt_len, ), generator=g).to(self.device)
        pos = torch.arange(prompt_len, device=self.device).repeat(b)
        blk = torch.gather(self.block_tables.long(), 1,
                           (torch.arange(prompt_len, device=self.device) // bs).expand(b, -1))
        slots = (blk * bs + (torch.arange(prompt_len, device=self.device) % bs)).view(-1)
        cu = torch.arange(0, (b + 1) * prompt_len, prompt_len, dtype=torch.int32, device=self.device)
        md = ROCmHipAttentionMetadata(
            num_prefills=b, num_prefill_tokens=b * prompt_len, num_decode_tokens=0,
            slot_mapping=slots, seq_lens=[prompt_len] * b,
            seq_lens_tensor=torch.full((b, ), prompt_len, dtype=torch.int32, device=self.device),
            max_query_len=prompt_len, max_prefill_seq_len=prompt_len, max_decode_seq_len=0,
            query_start_loc=cu, seq_start_loc=cu,
            context_lens_tensor=torch.zeros(b, dtype=torch.int32, device=self.device),
            block_tables=self.block_tables[:, :0], use_cuda_graph=False)
        hidden = self.model(ids, pos, self.kv_caches, md)
        last = hidden.view(b, prompt_len, -1)[:, -1]
        return self._sample(last)

    def weight_bytes_per_step(self) -> int:
        """bytes of parameters one decode step has to stream (this rank)"""
        total = 0
        for n, p in list(self.model.named_parameters()) + list(self.model.named_buffers()):
            if p.device.type == "meta" or n.endswith("g_idx") or "embed_tokens" in n or n.endswith("workspace"):
                continue
            if (n.endswith(".qweight") or n.endswith(".scales")) and (n.rsplit(".", 1)[0] + ".qweight_native") in self._native_names():
                continue   # a decode step reads the native copy of this layer, not a Marlin tensor kept beside it
            total += p.numel() * p.element_size()
        return total

    def _native_names(self):
        got = getattr(self, "_native_name_set", None)
        if got is None:
            got = self._native_name_set = {n for n, _ in self.model.named_buffers() if n.endswith("_native")}
        return got

    def resident_weight_bytes(self) -> int:
        """bytes of parameters and buffers this rank holds (after the first call: derived tensors built, released ones gone)"""
        seen, total = set(), 0
        for n, p in list(self.model.named_parameters()) + list(self.model.named_buffers()):
            if p.device.type == "meta" or p.data_ptr() in seen:
                continue
            seen.add(p.data_ptr())
            total += p.numel() * p.element_size()
        return total
