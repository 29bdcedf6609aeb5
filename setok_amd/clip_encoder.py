"""CLIPVisionTower — MI355X-native counterpart of the reference's frozen ViT wrapper
(src/model/setok/clip_encoder.py:8-93).

Same constructor arguments, attributes (`is_loaded`, `load_model`, `image_processor`, `select_layer`,
`select_feature`, `hidden_size`, `num_patches`, `num_patches_per_side`, `config`, `dtype`, `device`,
`dummy_feature`) and state-dict key names as the reference's `AutoModel`-backed tower, but the
arithmetic (HF CLIP ViT: patch conv, class/position embedding, pre-LayerNorm, pre-LN encoder layers
with quick_gelu MLP) runs on the HIP library.  Only the layers needed for `hidden_states[select_layer]`
are executed (select_layer=-2 skips the last encoder layer).
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace
from typing import Any, Dict, Optional

import torch
import torch.nn as nn

from . import autograd, ops
from ._packcache import PackCacheMixin

_VIT_DEFAULTS = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
                     image_size=224, patch_size=14, layer_norm_eps=1e-5, num_channels=3, hidden_act="quick_gelu")


def _as_config(spec: Any) -> SimpleNamespace:
    """Accepts a directory holding an HF `config.json`, a dict, or any object with the CLIP vision
    config attributes."""
    if isinstance(spec, str):
        path = os.path.join(spec, "config.json")
        if not os.path.isfile(path):
            raise OSError(f"{spec} is not a local directory with a config.json (there is no network; "
                          f"pass a local HF CLIP vision model directory or a config dict)")
        d = json.load(open(path))
        d = d.get("vision_config", d)
    elif isinstance(spec, dict):
        d = dict(spec)
    else:
        d = {k: getattr(spec, k) for k in _VIT_DEFAULTS if hasattr(spec, k)}
    cfg = dict(_VIT_DEFAULTS)
    cfg.update({k: v for k, v in d.items() if k in _VIT_DEFAULTS})
    if cfg["hidden_act"] != "quick_gelu":
        raise ValueError(f"unsupported CLIP hidden_act {cfg['hidden_act']!r} (quick_gelu only)")
    if cfg["num_channels"] != 3:
        raise ValueError("num_channels must be 3")
    return SimpleNamespace(**cfg)


class _ClipAttention(nn.Module):
    def __init__(self, C):
        super().__init__()
        self.k_proj, self.v_proj, self.q_proj, self.out_proj = (nn.Linear(C, C) for _ in range(4))


class _ClipMlp(nn.Module):
    def __init__(self, C, I):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(C, I), nn.Linear(I, C)


class _ClipLayer(nn.Module):
    def __init__(self, C, I, eps):
        super().__init__()
        self.self_attn = _ClipAttention(C)
        self.layer_norm1 = nn.LayerNorm(C, eps=eps)
        self.mlp = _ClipMlp(C, I)
        self.layer_norm2 = nn.LayerNorm(C, eps=eps)


class _ClipEmbeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        C, p = cfg.hidden_size, cfg.patch_size
        self.class_embedding = nn.Parameter(torch.randn(C) * C ** -0.5)
        self.patch_embedding = nn.Conv2d(3, C, kernel_size=p, stride=p, bias=False)
        self.position_embedding = nn.Embedding((cfg.image_size // p) ** 2 + 1, C)


class _ClipEncoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layers = nn.ModuleList(_ClipLayer(cfg.hidden_size, cfg.intermediate_size, cfg.layer_norm_eps)
                                    for _ in range(cfg.num_hidden_layers))


class ClipVisionParams(nn.Module):
    """Parameter container with HF `CLIPVisionModel` key names (transformers 5.x spelling
    `embeddings.*`, `encoder.layers.*`; the 4.x `vision_model.` prefix is stripped on load)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.embeddings = _ClipEmbeddings(cfg)
        self.pre_layrnorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.encoder = _ClipEncoder(cfg)
        self.post_layernorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)   # unused by hidden_states
        self._register_load_state_dict_pre_hook(self._strip_vision_model_prefix)

    @staticmethod
    def _strip_vision_model_prefix(state_dict, prefix, *args):
        old = prefix + "vision_model."
        for k in [k for k in state_dict if k.startswith(old)]:
            state_dict[prefix + k[len(old):]] = state_dict.pop(k)

    @property
    def dtype(self):
        return self.embeddings.class_embedding.dtype

    @property
    def device(self):
        return self.embeddings.class_embedding.device

    @property
    def config(self):
        return self.cfg


class CLIPVisionTower(PackCacheMixin, nn.Module):
    def __init__(self, vision_tower: Optional[Any], unfreeze_mm_vision_tower: Optional[bool] = False,
                 mm_vision_select_feature: Optional[str] = "patch", mm_vision_select_layer: Optional[int] = -2,
                 delay_load=False):
        super().__init__()
        self.is_loaded = False
        self.vision_tower_name = vision_tower
        self.select_layer = mm_vision_select_layer
        self.select_feature = mm_vision_select_feature
        self.image_processor = None
        self._init_pack_cache()
        if not delay_load or unfreeze_mm_vision_tower:                 # clip_encoder.py:22-27
            self.load_model()
        else:
            self.cfg_only = _as_config(vision_tower)

    # -- loading -------------------------------------------------------------------------------
    def load_model(self, device_map=None):
        if self.is_loaded:                                              # clip_encoder.py:30-32
            print("{} is already loaded, `load_model` called again, skipping.".format(self.vision_tower_name))
            return
        cfg = _as_config(self.vision_tower_name)
        self.vision_tower = ClipVisionParams(cfg)
        if isinstance(self.vision_tower_name, str):
            self._load_local_weights(self.vision_tower_name)
            self.image_processor = _load_image_processor(self.vision_tower_name)
        if device_map not in (None, "auto") and not isinstance(device_map, dict):
            self.vision_tower.to(device_map)
        self.vision_tower.requires_grad_(False)                         # clip_encoder.py:36
        self.is_loaded = True

    def _load_local_weights(self, path: str) -> None:
        st = os.path.join(path, "model.safetensors")
        pt = os.path.join(path, "pytorch_model.bin")
        if os.path.isfile(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        elif os.path.isfile(pt):
            sd = torch.load(pt, map_location="cpu")
        else:
            return                                                      # config only: random init
        sd = {k[len("vision_model."):] if k.startswith("vision_model.") else k: v for k, v in sd.items()}
        sd = {k: v for k, v in sd.items() if not k.startswith(("text_model.", "visual_projection", "text_projection", "logit_scale"))}
        self.vision_tower.load_state_dict(sd, strict=False)

    # -- cached compute-ready weights ------------------------------------------------------------
    def _pack(self):
        """Compute-ready views of the parameters: fused [q;k;v] projection, patch conv as a (C, Kpad)
        GEMM operand, fp32 biases / LayerNorm affine."""
        vt = self.vision_tower
        # one weight per encoder layer + the embeddings: a state-dict load rewrites all of them (and fires the post-hook anyway)
        # every parameter's counter (a few hundred integer reads per forward): an in-place update of ANY of them — a LoRA merge into q_proj / v_proj,
        # an edited layer_norm1 — must rebuild the fused q|k|v and folded-LayerNorm copies
        key = (vt.dtype, str(vt.device), os.environ.get("SETOK_LN_FOLD", "1"), self._versions(vt.parameters()))
        if self._packed.get("key") == key:
            return self._packed
        cfg, dt = vt.cfg, vt.dtype
        f32 = lambda t: t.detach().float().contiguous()
        C, p = cfg.hidden_size, cfg.patch_size
        kpad = ops.round_up(3 * p * p, 64)
        wp = torch.zeros((C, kpad), dtype=dt, device=vt.device)
        wp[:, :3 * p * p] = vt.embeddings.patch_embedding.weight.detach().reshape(C, -1)
        # bf16 throughput mode: layer_norm1 / layer_norm2 are folded into the q|k|v and fc1 GEMMs (ops.linear_ln); SETOK_LN_FOLD=0 keeps the
        # separate LayerNorm pass (A/B runs).  The fp32 parity mode always keeps it.
        fold = dt in ops.LOW and vt.device.type == "cuda" and os.environ.get("SETOK_LN_FOLD", "1") != "0" and C % 64 == 0 \
            and cfg.intermediate_size % 64 == 0
        layers = []
        for l in vt.encoder.layers:
            at = l.self_attn
            layers.append(dict(
                ln1=(f32(l.layer_norm1.weight), f32(l.layer_norm1.bias)),
                ln2=(f32(l.layer_norm2.weight), f32(l.layer_norm2.bias)),
                wqkv=torch.cat([at.q_proj.weight, at.k_proj.weight, at.v_proj.weight], 0).detach().contiguous(),
                bqkv=torch.cat([f32(at.q_proj.bias), f32(at.k_proj.bias), f32(at.v_proj.bias)], 0).contiguous(),
                wo=at.out_proj.weight.detach().contiguous(), bo=f32(at.out_proj.bias),
                w1=l.mlp.fc1.weight.detach().contiguous(), b1=f32(l.mlp.fc1.bias),
                w2=l.mlp.fc2.weight.detach().contiguous(), b2=f32(l.mlp.fc2.bias)))
            if fold:
                d = layers[-1]
                d["qkv_ln"] = ops.ln_fold(d["wqkv"], *d["ln1"], d["bqkv"])
                d["fc1_ln"] = ops.ln_fold(d["w1"], *d["ln2"], d["b1"])
        self._packed = dict(key=key, kpad=kpad, wp=wp, cls=vt.embeddings.class_embedding.detach().contiguous(),
                            pos=vt.embeddings.position_embedding.weight.detach().contiguous(),
                            pre=(f32(vt.pre_layrnorm.weight), f32(vt.pre_layrnorm.bias)), layers=layers)
        return self._packed

    # -- forward ---------------------------------------------------------------------------------
    def layers_needed(self) -> int:
        L = self.config.num_hidden_layers
        idx = self.select_layer if self.select_layer >= 0 else L + 1 + self.select_layer
        if not 0 <= idx <= L:
            raise IndexError(f"select_layer {self.select_layer} out of range")
        return idx

    def hidden_rows(self, images: torch.Tensor) -> torch.Tensor:
        """(B*(N+1), C) rows of hidden_states[select_layer] (class token first in each image).  Inference only, like the reference's
        `@torch.no_grad()` forward (clip_encoder.py:50): a tower whose parameters were unfrozen (`unfreeze_mm_vision_tower`, or a manual
        `requires_grad_(True)`) gets no gradient in the reference either; here that is said once, as a warning."""
        if not self.is_loaded:
            raise RuntimeError("vision tower not loaded: call load_model() first")
        autograd.warn_no_grad_once("CLIPVisionTower", self.vision_tower.parameters(),
                                   "the tower's forward is @torch.no_grad() in the reference (clip_encoder.py:50) and has no backward pass on the HIP path",
                                   owner=self, inputs=[images])
        with torch.no_grad():
            return self._hidden_rows(images)

    def _hidden_rows(self, images: torch.Tensor) -> torch.Tensor:
        cfg = self.config
        if images.dim() != 4 or images.shape[1] != 3 or images.shape[2] != cfg.image_size or images.shape[3] != cfg.image_size:
            raise ValueError(f"Input image size ({images.shape[-2]}*{images.shape[-1]}) doesn't match model "
                             f"({cfg.image_size}*{cfg.image_size}).")
        pk = self._pack()
        x = images.to(device=self.device, dtype=self.dtype).contiguous()
        B, C, H = x.shape[0], cfg.hidden_size, cfg.num_attention_heads
        N, T, Dh, eps = (cfg.image_size // cfg.patch_size) ** 2, (cfg.image_size // cfg.patch_size) ** 2 + 1, C // H, cfg.layer_norm_eps
        patches = ops.patchify(x, cfg.patch_size, pk["kpad"])
        pe = ops.linear(patches, pk["wp"])
        h = ops.vit_assemble(pe, pk["cls"], pk["pos"], B, N)
        ops.layernorm(h, *pk["pre"], eps, out=h)
        st = y = None
        for l in pk["layers"][: self.layers_needed()]:
            if "qkv_ln" in l:                          # LayerNorm folded into the consumer: statistics only, the GEMM reads the raw stream
                st = ops.row_stats(h, eps, out=st)
                qkv = ops.linear_ln(h, l["qkv_ln"], st)
            else:
                y = ops.layernorm(h, *l["ln1"], eps, out=y)
                qkv = ops.linear(y, l["wqkv"], l["bqkv"])
            a = ops.attention(qkv, H, Dh, Dh ** -0.5, seg_len=T)
            ops.linear(a, l["wo"], l["bo"], residual=h, out=h)
            if "fc1_ln" in l:
                st = ops.row_stats(h, eps, out=st)
                u = ops.linear_ln(h, l["fc1_ln"], st, act=ops.ACT_QUICK_GELU)
            else:
                y = ops.layernorm(h, *l["ln2"], eps, out=y)
                u = ops.linear(y, l["w1"], l["b1"], act=ops.ACT_QUICK_GELU)
            ops.linear(u, l["w2"], l["b2"], residual=h, out=h)
        return h

    def feature_select(self, hidden_rows: torch.Tensor, B: int) -> torch.Tensor:
        C = hidden_rows.shape[-1]
        feats = hidden_rows.reshape(B, -1, C)
        if self.select_feature == "patch":                              # clip_encoder.py:42-47
            return feats[:, 1:]
        if self.select_feature == "cls_patch":
            return feats
        raise ValueError(f"Unexpected select feature: {self.select_feature}")

    @torch.no_grad()
    def forward(self, images):
        if type(images) is list:                                        # clip_encoder.py:52-57
            return [self.feature_select(self.hidden_rows(im.unsqueeze(0)), 1).to(im.dtype) for im in images]
        return self.feature_select(self.hidden_rows(images), images.shape[0]).to(images.dtype)

    # -- properties (clip_encoder.py:64-93) --------------------------------------------------------
    @property
    def dummy_feature(self):
        return torch.zeros(1, self.hidden_size, device=self.device, dtype=self.dtype)

    @property
    def dtype(self):
        return self.vision_tower.dtype

    @property
    def device(self):
        return self.vision_tower.device

    @property
    def config(self):
        return self.vision_tower.config if self.is_loaded else self.cfg_only

    @property
    def hidden_size(self):
        return self.config.hidden_size

    @property
    def num_patches_per_side(self):
        return self.config.image_size // self.config.patch_size

    @property
    def num_patches(self):
        return (self.config.image_size // self.config.patch_size) ** 2


def _load_image_processor(path: str):
    """`AutoProcessor.from_pretrained(path)` (clip_encoder.py:34) when a preprocessor config is present."""
    if not os.path.isfile(os.path.join(path, "preprocessor_config.json")):
        return None
    try:
        from transformers import AutoProcessor
        return AutoProcessor.from_pretrained(path)
    except Exception:       # host-side convenience only; never on the compute path
        return None
