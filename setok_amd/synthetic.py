"""Seeded synthetic weights for benchmarking (no pretrained weights / network exist on the GPU box).

Standard deviations follow HF CLIP's published initialisation so that token features keep a
realistic spread through the depth — with it the DPC-kNN scores of random images land around
0.105-0.155 and `threshold=0.125` makes the dynamic-k branch fire."""
from __future__ import annotations

import torch

from .tokenizer import SetokTokenizer


@torch.no_grad()
def init_synthetic_(tok: SetokTokenizer, tower_seed: int = 0, head_seed: int = 1) -> SetokTokenizer:
    vt = tok.image_feature_encoder.vision_tower
    cfg = vt.cfg
    C, L = cfg.hidden_size, cfg.num_hidden_layers
    g = torch.Generator().manual_seed(tower_seed)
    in_std, out_std, fc_std = C ** -0.5 * (2 * L) ** -0.5, C ** -0.5, (2 * C) ** -0.5

    def fill(p, std):
        p.copy_(torch.randn(p.shape, generator=g, dtype=torch.float32) * std)

    fill(vt.embeddings.class_embedding, C ** -0.5)
    fill(vt.embeddings.patch_embedding.weight, 0.02)
    fill(vt.embeddings.position_embedding.weight, 0.02)
    for ln in (vt.pre_layrnorm,):
        ln.weight.fill_(1.0); ln.bias.zero_()
    for l in vt.encoder.layers:
        for nm in ("q_proj", "k_proj", "v_proj"):
            fill(getattr(l.self_attn, nm).weight, in_std); getattr(l.self_attn, nm).bias.zero_()
        fill(l.self_attn.out_proj.weight, out_std); l.self_attn.out_proj.bias.zero_()
        fill(l.mlp.fc1.weight, fc_std); l.mlp.fc1.bias.zero_()
        fill(l.mlp.fc2.weight, in_std); l.mlp.fc2.bias.zero_()
        for ln in (l.layer_norm1, l.layer_norm2):
            ln.weight.fill_(1.0); ln.bias.zero_()
    torch.manual_seed(head_seed)
    tok.inner_encoder.apply(tok._init_weights)
    tok.inter_encoder.apply(tok._init_weights)
    tok._init_weights(tok.out)
    return tok
