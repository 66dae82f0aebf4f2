"""Plain-torch fp32 reference of the Llama decoder used to check end-to-end logits (test infrastructure; runs on
whatever device its weights are on -- the CPU for the small models, the GPU for the full-size single layer).  Mirrors what the reference's CPU executor computes for
vllm/model_executor/models/llama.py: RMSNorm -> qkv -> neox rope -> causal attention over the
whole sequence -> o_proj -> RMSNorm -> silu(gate)*up -> down_proj, with dequantised GPTQ weights."""
from typing import Dict

import torch


def dequant_gptq(qweight, scales, bits=4):
    pf = 32 // bits
    k = qweight.shape[0] * pf
    q = qweight.to(torch.int64) & 0xFFFFFFFF
    shifts = (torch.arange(pf, dtype=torch.int64, device=qweight.device) * bits).view(1, pf, 1)
    codes = ((q[:, None, :] >> shifts) & (2**bits - 1)).reshape(k, -1).float()
    g = k // scales.shape[0]
    w = (codes - 2**(bits - 1)) * scales.float().repeat_interleave(g, dim=0)
    return w.to(scales.dtype).float()  # the kernel semantics: (q - 8) * s, rounded to the model dtype


class RefLlama:

    def __init__(self, arch, weights: Dict[str, torch.Tensor], act_int8: bool = False):
        self.a = arch
        self.act_int8 = act_int8
        self.w = {}
        for name, t in weights.items():
            self.w[name] = t
        self.lin = {}

    def linear(self, prefix, x):
        if prefix not in self.lin:
            if prefix + ".qweight" in self.w:
                self.lin[prefix] = dequant_gptq(self.w[prefix + ".qweight"], self.w[prefix + ".scales"])
            elif prefix + ".weight_scale" in self.w:
                # compressed-tensors W8A8: int8 [out, in] x per-channel fp32 scale (weights exact)
                self.lin[prefix] = (self.w[prefix + ".weight"].float() * self.w[prefix + ".weight_scale"]).t()
            else:
                self.lin[prefix] = self.w[prefix + ".weight"].float().t()
        if self.act_int8 and prefix + ".weight_scale" in self.w:
            # dynamic per-token int8 activations (int8_quant_kernels.cu:39-75): x ~ round(x/s)*s
            s = x.abs().amax(dim=-1, keepdim=True).clamp_min(1e-12) / 127.0
            x = torch.clamp(torch.round(x / s), -128, 127) * s
        return x @ self.lin[prefix]

    def rms(self, x, w):
        v = x.pow(2).mean(-1, keepdim=True)
        return x * torch.rsqrt(v + self.a.rms_norm_eps) * w.float()

    def rope(self, x, pos):
        hd = self.a.head_dim
        inv = 1.0 / (self.a.rope_theta**(torch.arange(0, hd, 2, device=x.device).float() / hd))
        f = pos.float()[:, None] * inv[None]
        cos, sin = f.cos()[:, None, :], f.sin()[:, None, :]
        x1, x2 = x[..., :hd // 2], x[..., hd // 2:]
        return torch.cat((x1 * cos - x2 * sin, x2 * cos + x1 * sin), dim=-1)

    def forward(self, ids: torch.Tensor) -> torch.Tensor:
        """ids [L] -> logits [L, vocab] (causal)"""
        a = self.a
        L = ids.shape[0]
        pos = torch.arange(L, device=ids.device)
        h = self.w["model.embed_tokens.weight"].float()[ids]
        for i in range(a.num_hidden_layers):
            p = f"model.layers.{i}."
            x = self.rms(h, self.w[p + "input_layernorm.weight"])
            q = self.linear(p + "self_attn.q_proj", x).view(L, a.num_attention_heads, a.head_dim)
            k = self.linear(p + "self_attn.k_proj", x).view(L, a.num_key_value_heads, a.head_dim)
            v = self.linear(p + "self_attn.v_proj", x).view(L, a.num_key_value_heads, a.head_dim)
            q, k = self.rope(q, pos), self.rope(k, pos)
            rep = a.num_attention_heads // a.num_key_value_heads
            k, v = k.repeat_interleave(rep, 1), v.repeat_interleave(rep, 1)
            att = torch.einsum("qhd,khd->hqk", q, k) * a.head_dim**-0.5
            att = att + torch.full((L, L), float("-inf"), device=ids.device).triu(1)[None]
            o = torch.einsum("hqk,khd->qhd", att.softmax(-1), v).reshape(L, -1)
            h = h + self.linear(p + "self_attn.o_proj", o)
            x = self.rms(h, self.w[p + "post_attention_layernorm.weight"])
            g = self.linear(p + "mlp.gate_proj", x)
            u = self.linear(p + "mlp.up_proj", x)
            h = h + self.linear(p + "mlp.down_proj", torch.nn.functional.silu(g) * u)
        h = self.rms(h, self.w["model.norm.weight"])
        return h @ self.w["lm_head.weight"].float().t()
