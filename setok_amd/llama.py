"""LLM prefill of BASELINE config 5 on the HIP library: what `SetokimLlamaForCausalLM.forward`
(src/model/language_model/setokim_llama.py:94-143) runs after `prepare_inputs_labels_for_multimodal` —

    outputs = self.model(attention_mask=..., position_ids=..., inputs_embeds=...)      # HF LlamaModel      :130-139
    logits  = self.lm_head(outputs[0])                                                 #                    :143

`self.model` is HuggingFace `transformers` LlamaModel (third-party code; Vicuna-7B = 32 layers, hidden 4096, 32 heads of 128, SwiGLU
11008, RMSNorm, rotary embeddings).  This module holds the same parameters under the same state-dict names
(`model.embed_tokens.weight`, `model.layers.{i}.self_attn.{q,k,v,o}_proj.weight`, `model.layers.{i}.mlp.{gate,up,down}_proj.weight`,
`model.layers.{i}.{input,post_attention}_layernorm.weight`, `model.norm.weight`, `lm_head.weight`) so a Vicuna / Llama-2 checkpoint loads
unchanged, and runs the prefill (no KV cache, eager-attention arithmetic) as: RMSNorm -> one fused q|k|v GEMM -> rotary embedding in place
-> causal + padding-masked MFMA attention -> o_proj GEMM with the residual fused -> RMSNorm -> one fused gate|up GEMM -> SwiGLU ->
down_proj GEMM with the residual fused.  Inference only.  Grouped-query attention (num_key_value_heads < num_attention_heads: Llama-2-70B,
Llama-3, Mistral — Vicuna-7B / 13B do not use it) runs on the same kernels: the fused q|k|v rows are [H | Hkv | Hkv] heads wide and query head h
reads key / value head h // (H // Hkv), HF's `repeat_kv` without the copies.
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch
import torch.nn as nn

from . import autograd, ops
from ._packcache import PackCacheMixin
from .arch import SetokimVisionMixin, config_get


class _RMSNorm(nn.Module):
    def __init__(self, dim, eps):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.variance_epsilon = eps


class _Attention(nn.Module):
    def __init__(self, D, H, Hkv):
        super().__init__()
        dh = D // H
        self.q_proj = nn.Linear(D, H * dh, bias=False)
        self.k_proj = nn.Linear(D, Hkv * dh, bias=False)
        self.v_proj = nn.Linear(D, Hkv * dh, bias=False)
        self.o_proj = nn.Linear(H * dh, D, bias=False)


class _MLP(nn.Module):
    def __init__(self, D, F):
        super().__init__()
        self.gate_proj = nn.Linear(D, F, bias=False)
        self.up_proj = nn.Linear(D, F, bias=False)
        self.down_proj = nn.Linear(F, D, bias=False)


class _DecoderLayer(nn.Module):
    def __init__(self, D, H, Hkv, F, eps):
        super().__init__()
        self.self_attn = _Attention(D, H, Hkv)
        self.mlp = _MLP(D, F)
        self.input_layernorm = _RMSNorm(D, eps)
        self.post_attention_layernorm = _RMSNorm(D, eps)


class LlamaModel(PackCacheMixin, nn.Module):
    """Parameter container with HF LlamaModel's tree + the prefill on the HIP library."""

    def __init__(self, vocab_size, hidden_size, intermediate_size, num_hidden_layers, num_attention_heads, num_key_value_heads, rms_norm_eps,
                 rope_theta):
        super().__init__()
        if num_key_value_heads < 1 or num_attention_heads % num_key_value_heads != 0:
            raise ValueError(f"num_attention_heads ({num_attention_heads}) must be a multiple of num_key_value_heads ({num_key_value_heads})")
        self.num_heads, self.head_dim, self.rope_theta, self.eps = num_attention_heads, hidden_size // num_attention_heads, rope_theta, rms_norm_eps
        self.num_kv_heads = num_key_value_heads
        self.embed_tokens = nn.Embedding(vocab_size, hidden_size)
        self.layers = nn.ModuleList([_DecoderLayer(hidden_size, num_attention_heads, num_key_value_heads, intermediate_size, rms_norm_eps)
                                     for _ in range(num_hidden_layers)])
        self.norm = _RMSNorm(hidden_size, rms_norm_eps)
        self._init_pack_cache()

    def _pack(self):
        w = self.norm.weight
        key = (w.dtype, str(w.device), self._versions(self.parameters()))      # every tensor: a LoRA merge into q_proj / v_proj alone must rebuild the fused copies
        if self._packed.get("key") == key:
            return self._packed
        f32 = lambda t: t.detach().float().contiguous()
        layers = []
        for l in self.layers:
            a, m = l.self_attn, l.mlp
            layers.append(dict(n1=f32(l.input_layernorm.weight), n2=f32(l.post_attention_layernorm.weight),
                               wqkv=torch.cat([a.q_proj.weight.detach(), a.k_proj.weight.detach(), a.v_proj.weight.detach()], 0).contiguous(),
                               wo=a.o_proj.weight.detach().contiguous(),
                               wgu=ops.interleave_gate_up(m.gate_proj.weight.detach(), m.up_proj.weight.detach()),      # (gate_j, up_j) row pairs: linear_swiglu
                               wd=m.down_proj.weight.detach().contiguous()))
        self._packed = dict(key=key, layers=layers, norm=f32(self.norm.weight))
        return self._packed

    def forward(self, inputs_embeds: torch.Tensor, attention_mask: Optional[torch.Tensor] = None, position_ids: Optional[torch.Tensor] = None):
        """inputs_embeds (B, T, D); attention_mask (B, T), 1 = token (None = all); position_ids (B, T) (None = 0..T-1).
        Returns the hidden states after the final norm, (B, T, D).  A prefill: inference arithmetic, no autograd graph (where torch would have
        recorded one the result's backward raises — training the LLM is the reference's HF-Trainer side, SURVEY.md §2 "OUT")."""
        with torch.no_grad():
            out = self._forward(inputs_embeds, attention_mask, position_ids)
        return autograd.no_backward("LlamaModel.forward (the prefill has no backward pass on the HIP path)", out, [inputs_embeds, *self.parameters()])

    def _forward(self, inputs_embeds, attention_mask=None, position_ids=None):
        B, T, D = inputs_embeds.shape
        pk = self._pack()
        H, Hkv, dh = self.num_heads, self.num_kv_heads, self.head_dim
        dev = inputs_embeds.device
        x = inputs_embeds.to(self.norm.weight.dtype).reshape(B * T, D).contiguous().clone()
        if position_ids is None:
            position_ids = torch.arange(T, device=dev)[None].expand(B, T)
        pos = position_ids.to(device=dev, dtype=torch.int64).reshape(B * T).contiguous()
        km = None if attention_mask is None else attention_mask.to(device=dev).bool().to(torch.uint8).reshape(B * T).contiguous()
        y = None
        for L in pk["layers"]:
            y = ops.rmsnorm(x, L["n1"], self.eps, out=y)
            qkv = ops.linear(y, L["wqkv"])
            ops.rope_(qkv, pos, H, dh, self.rope_theta, Hkv)
            o = ops.attention_causal(qkv, km, B, T, H, dh, dh ** -0.5, Hkv)
            ops.linear(o, L["wo"], residual=x, out=x)
            y = ops.rmsnorm(x, L["n2"], self.eps, out=y)
            g = ops.linear_swiglu(y, L["wgu"])                           # gate|up Linear with SwiGLU in its epilogue (16-bit modes; the unfused pair otherwise: same bits)
            ops.linear(g, L["wd"], residual=x, out=x)
        return ops.rmsnorm(x, pk["norm"], self.eps, out=y).reshape(B, T, D)


def _refuse_unsupported_llama_fields(g) -> None:
    """The prefill reads hidden / intermediate sizes, layer and head counts, `num_key_value_heads`, `rms_norm_eps` and `rope_theta` — plain
    multi-head or grouped-query Llama (Vicuna-7B, the reference's LLM: scripts/finetune.sh).  A checkpoint whose config carries anything
    that changes the arithmetic beyond that would load cleanly and give wrong logits, so it is refused here:
    `rope_scaling` (llama3 / linear / dynamic / yarn: other inverse frequencies at every position), an explicit `head_dim` that is not
    hidden_size / num_attention_heads, `attention_bias` / `mlp_bias` (the fused q|k|v, gate|up and down projections have no bias input),
    and a `sliding_window` (checked against the sequence length in forward: a window that covers the prompt is plain causal attention)."""
    for key in ("rope_scaling", "rope_parameters"):                         # (`rope_parameters`: the name newer HF configs carry the same dict under)
        rs = g(key)
        if rs is None:
            continue
        kind = (rs.get("rope_type", rs.get("type")) if isinstance(rs, dict) else getattr(rs, "rope_type", getattr(rs, "type", rs)))
        if kind not in (None, "default"):
            raise NotImplementedError(f"SetokimLlamaPrefill: {key}={rs!r} is not implemented on the HIP path (only the plain rotary "
                                      "embedding with `rope_theta`): a Llama-3.1-style checkpoint would give wrong logits")
    _rope_theta(g)                                                          # (raises on two different values)
    hd, H, D = g("head_dim"), g("num_attention_heads"), g("hidden_size")
    if hd is not None and int(hd) != int(D) // int(H):
        raise NotImplementedError(f"SetokimLlamaPrefill: head_dim={hd} != hidden_size / num_attention_heads = {int(D) // int(H)} is not implemented")
    for k in ("attention_bias", "mlp_bias"):
        if g(k, False):
            raise NotImplementedError(f"SetokimLlamaPrefill: {k}=True is not implemented (the projections are loaded without a bias)")


def _rope_theta(g) -> float:
    """The rotary base.  Older HF configs carry it as the top-level `rope_theta`; newer ones keep it INSIDE `rope_parameters` (and some inside
    `rope_scaling`) next to `rope_type: "default"`, which the guard above accepts — reading only the top-level field then silently used 10000
    for a 500000-base checkpoint.  The dict's value wins when the top level has none; two different values are refused."""
    top = g("rope_theta")
    inner = None
    for key in ("rope_parameters", "rope_scaling"):
        rs = g(key)
        if rs is None:
            continue
        v = rs.get("rope_theta") if isinstance(rs, dict) else getattr(rs, "rope_theta", None)
        if v is None:
            continue
        if inner is not None and float(v) != float(inner):
            raise NotImplementedError(f"SetokimLlamaPrefill: rope_parameters / rope_scaling carry two different rope_theta values ({inner} and {v})")
        inner = v
    if top is not None and inner is not None and float(top) != float(inner):
        raise NotImplementedError(f"SetokimLlamaPrefill: rope_theta={top} at the top level of the config but {inner} inside rope_parameters: "
                                  "refusing to pick one")
    if inner is not None:
        return float(inner)
    return float(top) if top is not None else 10000.0


class SetokimLlamaPrefill(nn.Module, SetokimVisionMixin):
    """`SetokimLlamaForCausalLM.forward` without the loss (setokim_llama.py:94-143): splice the image tokens into the text embeddings,
    run the LLM over them, project to the vocabulary.  `vision_tower` / `mm_in_projector` are the SetokTokenizer and projector of the
    encode path (may be None when `inputs_embeds` are passed directly)."""

    def __init__(self, config: Any, vision_tower=None, mm_in_projector=None):
        super().__init__()
        g = lambda k, d=None: config_get(config, k, d)
        self.config = config
        _refuse_unsupported_llama_fields(g)
        self.model = LlamaModel(g("vocab_size"), g("hidden_size"), g("intermediate_size"), g("num_hidden_layers"), g("num_attention_heads"),
                                g("num_key_value_heads", g("num_attention_heads")), g("rms_norm_eps", 1e-5), _rope_theta(g))
        self.lm_head = nn.Linear(g("hidden_size"), g("vocab_size"), bias=False)
        self.vision_tower = vision_tower
        self.mm_in_projector = mm_in_projector

    def get_model(self):
        return self.model

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, inputs_embeds=None, labels=None, comp_images=None,
                last_token_only: bool = False, return_loss: bool = False):
        with torch.no_grad():
            out = self._forward(input_ids, attention_mask, position_ids, inputs_embeds, labels, comp_images, last_token_only, return_loss)
        return autograd.no_backward("SetokimLlamaPrefill.forward (inference: encode -> splice -> prefill -> logits)", out, [inputs_embeds, *self.parameters()])

    def _forward(self, input_ids=None, attention_mask=None, position_ids=None, inputs_embeds=None, labels=None, comp_images=None,
                 last_token_only: bool = False, return_loss: bool = False):
        """Returns (logits, new_labels, attention_mask): logits (B, T', vocab) — or (B, vocab) for every sequence's last token with
        `last_token_only` (what a generation step after the prefill needs) — on the spliced sequence of length T'.  With `return_loss`
        (and labels) a fourth element is the language-model loss of setokim_llama.py:145-160 (shifted cross entropy over the positions whose
        next token is neither padding nor IGNORE_INDEX), as a 0-d fp32 tensor; the diffusion term of :163-180 is not part of this path."""
        new_labels = labels
        self._last_features = None
        if inputs_embeds is None:
            _, position_ids, attention_mask, _, inputs_embeds, new_labels = self.prepare_inputs_labels_for_multimodal(
                input_ids, position_ids, attention_mask, None, labels, comp_images)
            if inputs_embeds is None:                                              # no images: plain text
                # text only: embed_tokens(input_ids).  An id the table cannot serve — an IMAGE_TOKEN_INDEX although no images came, a leaked
                # TARGET_TOKEN_INDEX, an id >= vocab — raises the reference's IndexError (torch's embedding) instead of becoming an address
                w_e = self.model.embed_tokens.weight.detach().contiguous()
                status = torch.empty(2, dtype=torch.int32, device=w_e.device)
                inputs_embeds = ops.splice_rows(input_ids.to(device=w_e.device, dtype=torch.int32).contiguous(), w_e, None, status)
                st = status.cpu()
                if int(st[0]) != 0:
                    b, t = divmod(int(st[1]), input_ids.shape[1])
                    raise IndexError(f"index out of range in self: input_ids[{b}, {t}] = {int(input_ids[b, t])} is not a row of the "
                                     f"{w_e.shape[0]}-row embedding table (and no images were passed for image placeholders)")
        sw = config_get(self.config, "sliding_window")
        if sw is not None and int(sw) < inputs_embeds.shape[1]:
            raise NotImplementedError(f"SetokimLlamaPrefill: sliding_window={sw} is shorter than the sequence ({inputs_embeds.shape[1]} positions): "
                                      "windowed attention is not implemented on the HIP path")
        hidden = self.model(inputs_embeds, attention_mask, position_ids)           # setokim_llama.py:130-140
        B, T, D = hidden.shape
        w = self.lm_head.weight.detach().contiguous()
        if last_token_only:
            if attention_mask is None:
                last = torch.full((B,), T - 1, device=hidden.device)
            else:
                am = attention_mask.to(hidden.device).bool()
                last = (am * torch.arange(T, device=hidden.device)[None]).max(dim=1).values
            rows = hidden[torch.arange(B, device=hidden.device), last].contiguous()
            return ops.linear(rows, w), new_labels, attention_mask
        logits = ops.linear(hidden.reshape(B * T, D), w).reshape(B, T, -1)         # :143
        if return_loss:
            if new_labels is None:
                raise ValueError("return_loss needs labels")
            return logits, new_labels, attention_mask, ops.lm_loss(logits, new_labels, attention_mask)[0]      # :145-160
        return logits, new_labels, attention_mask
