"""Factories the LLaVA-derived pipeline calls: `build_vision_tower`
(src/model/multimodal_encoder/builder.py:6-22), `build_vision_projector`
(src/model/multimodal_projector/builder.py:33-64) and `build_vision_generator`
(src/model/multimodal_generator/builder.py:4-12)."""
from __future__ import annotations

import re
from dataclasses import asdict, is_dataclass

import torch
import torch.nn as nn

from . import autograd
from .detokenizer import SetokDeTokenizer
from .tokenizer import SetokTokenizer


def build_vision_tower(vision_tower_cfg, **kwargs):
    """Same contract as the reference: `vision_tower_cfg` is a dataclass / dict / namespace holding the
    SetokTokenizer ctor kwargs; the tower spec is read from `vision_tokenizer` or `vision_tower`.

    The reference only accepts tower names containing 'siglip' (builder.py:19) although only a
    CLS-bearing CLIP ViT is shape-consistent with its feature_select; here a name
    containing 'siglip' or 'clip', a local HF directory, or a config dict/object is accepted, and
    anything else raises the reference's ValueError."""
    if is_dataclass(vision_tower_cfg):
        cfg = asdict(vision_tower_cfg)
    elif isinstance(vision_tower_cfg, dict):
        cfg = dict(vision_tower_cfg)
    else:
        cfg = dict(vars(vision_tower_cfg))
    vision_tower = cfg.get("vision_tokenizer", cfg.get("vision_tower", None))
    ok = (isinstance(vision_tower, str) and ("siglip" in vision_tower.lower() or "clip" in vision_tower.lower())) \
        or isinstance(vision_tower, dict) or (vision_tower is not None and hasattr(vision_tower, "hidden_size"))
    if not ok:
        raise ValueError(f"Unknown vision tower: {vision_tower}")
    cfg.pop("vision_tokenizer", None)
    cfg["vision_tower"] = vision_tower
    return SetokTokenizer(**cfg, **kwargs)


def build_vision_generator(image_generator_cfg, **kwargs):
    """multimodal_generator/builder.py:4-12: dataclass / dict / namespace of SetokDeTokenizer ctor kwargs."""
    if is_dataclass(image_generator_cfg):
        cfg = asdict(image_generator_cfg)
    elif isinstance(image_generator_cfg, dict):
        cfg = dict(image_generator_cfg)
    else:
        cfg = dict(vars(image_generator_cfg))
    return SetokDeTokenizer(**cfg, **kwargs)


class IdentityMap(nn.Module):                         # multimodal_projector/builder.py:6-15
    def forward(self, x, *args, **kwargs):
        return x

    @property
    def config(self):
        return {"mm_projector_type": "identity"}


class VisionProjector(nn.Sequential):
    """nn.Sequential-compatible parameter tree (keys `0.weight`, `2.weight`, ...) whose forward runs on
    the HIP library: Linear+GELU pairs are one GEMM with a fused exact-erf GELU epilogue.

    Differentiable like the reference's nn.Sequential (stage 2 trains exactly this module through the LLM loss,
    scripts/pretrain_mm_proj.sh:40): with gradients enabled and a parameter or the input requiring one, the same calls run inside
    `autograd.ProjectorFn` (forward bits unchanged) and `loss.backward()` fills `.grad` of the parameters and of the tokens."""

    def forward(self, x):
        if hasattr(x, "map") and hasattr(x, "packed"):            # RaggedTokens: project all tokens at once
            return x.map(self.forward)
        shape = x.shape
        h = x.reshape(-1, shape[-1])
        plan, tensors = autograd.projector_plan(list(self))
        if autograd.grad_needed(h, *tensors):
            h = autograd.ProjectorFn.apply(h, plan, self, *tensors)
        else:
            with torch.no_grad():
                h = autograd.ProjectorFn.forward(_NoCtx(), h.detach(), plan, self, *tensors)
        return h.reshape(*shape[:-1], h.shape[-1])


class _NoCtx:
    """Stand-in for the autograd context when no graph is recorded: the forward is one code path either way."""


def build_vision_projector(projector_type="linear", mm_hidden_size=4096, hidden_size=3078, delay_load=False, **kwargs):
    if projector_type == "linear":
        return VisionProjector(nn.Linear(mm_hidden_size, hidden_size))._as_linear()
    use_norm = False
    if "_Norm" in projector_type:
        use_norm = True
        projector_type = projector_type.replace("_Norm", "")
    mlp_gelu_match = re.match(r"^mlp(\d+)x_gelu$", projector_type)
    if mlp_gelu_match:
        mlp_depth = int(mlp_gelu_match.group(1))
        modules = [nn.Linear(mm_hidden_size, hidden_size)]
        if use_norm:
            modules.append(nn.LayerNorm(hidden_size))
        for _ in range(1, mlp_depth):
            modules.append(nn.GELU())
            modules.append(nn.Linear(hidden_size, hidden_size))
        return VisionProjector(*modules)
    if projector_type == "identity":
        return IdentityMap()
    raise ValueError(f"Unknown projector type: {projector_type}")


class LinearProjector(nn.Linear):
    """`projector_type == 'linear'`: a bare nn.Linear in the reference (keys `weight`, `bias`).  Differentiable like VisionProjector."""

    def forward(self, x):
        if hasattr(x, "map") and hasattr(x, "packed"):
            return x.map(self.forward)
        shape = x.shape
        h = x.reshape(-1, shape[-1])
        plan, tensors = (("linear", False, self.bias is not None),), [self.weight] + ([self.bias] if self.bias is not None else [])
        if autograd.grad_needed(h, *tensors):
            h = autograd.ProjectorFn.apply(h, plan, self, *tensors)
        else:
            with torch.no_grad():
                h = autograd.ProjectorFn.forward(_NoCtx(), h.detach(), plan, self, *tensors)
        return h.reshape(*shape[:-1], h.shape[-1])


def _as_linear(self):
    lin = self[0]
    out = LinearProjector(lin.in_features, lin.out_features)
    out.load_state_dict(lin.state_dict())
    return out


VisionProjector._as_linear = _as_linear
