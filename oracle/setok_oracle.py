"""CPU oracle for the SeTok `encode_images` hot path.  TEST INFRASTRUCTURE — NOT A PRODUCT PATH.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this module; the
product (setok_amd/) never does, and fails loudly when the HIP library is missing.

This is a from-scratch restatement (torch-CPU as the array library, fp32 by default, fp64 for
decision-margin analysis) of the arithmetic of the reference's path.  Every function cites the
reference file:line it follows (paths relative to /root/reference/).  It is pinned against the
reference itself ("RAC", oracle/rac_harness.py) by tests/test_oracle_vs_reference.py in the build
container, and against the committed golden vectors in tests/golden/ everywhere.

Third-party arithmetic on the path: the ViT tower is HuggingFace `transformers`
CLIPVisionModel (pinned transformers==4.46.3 in the reference's pyproject.toml:18, 5.15.0 installed
here; eager attention path, same math).  It is restated in `clip_vit_forward` from the published
CLIP ViT algorithm and pinned against the installed HF implementation through the reference's own
call site (src/model/setok/clip_encoder.py:50-62).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
TOWER_PREFIX = "image_feature_encoder.vision_tower."


# ----------------------------------------------------------------------------------------------
# configuration
# ----------------------------------------------------------------------------------------------
@dataclass
class VitConfig:
    """Subset of HF CLIPVisionConfig the tower arithmetic depends on."""
    hidden_size: int = 1024
    intermediate_size: int = 4096
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    image_size: int = 224
    patch_size: int = 14
    layer_norm_eps: float = 1e-5
    num_channels: int = 3

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def num_patches(self) -> int:
        return self.grid * self.grid


@dataclass
class HeadConfig:
    """ctor kwargs of SetokTokenizer (src/model/setok/tokenizer.py:14-34) that affect arithmetic."""
    hidden_dim: int = 1024
    token_feat_dim: int = 4096
    min_cluster_num: int = 64
    threshold: float = 0.5
    nheads: int = 2
    dim_feedforward: int = 4096
    inner_cluster_layers: int = 2
    intra_cluster_layers: int = 2
    mm_vision_select_layer: int = -2
    mm_vision_select_feature: str = "patch"


def normalise_tower_keys(sd: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """transformers 4.46 names tower params `vision_tower.vision_model.*`, 5.x `vision_tower.*`."""
    out = {}
    for k, v in sd.items():
        if k.startswith(TOWER_PREFIX + "vision_model."):
            k = TOWER_PREFIX + k[len(TOWER_PREFIX + "vision_model."):]
        out[k] = v
    return out


# ----------------------------------------------------------------------------------------------
# a1 — the ViT tower (third-party HF CLIP arithmetic; call site clip_encoder.py:50-62)
# ----------------------------------------------------------------------------------------------
def quick_gelu(x: Tensor) -> Tensor:
    """HF `quick_gelu` activation of CLIP's MLP: x * sigmoid(1.702 x)."""
    return x * torch.sigmoid(1.702 * x)


def clip_vit_hidden_states(sd: Dict[str, Tensor], cfg: VitConfig, images: Tensor,
                           n_layers: Optional[int] = None) -> List[Tensor]:
    """HF CLIPVisionModel(images, output_hidden_states=True).hidden_states, restated.

    hidden_states[0] is the embedding output *after* pre_layrnorm, hidden_states[i] the output of
    encoder layer i (post_layernorm is never applied to hidden states)."""
    p = TOWER_PREFIX
    C, H, dh = cfg.hidden_size, cfg.num_attention_heads, cfg.hidden_size // cfg.num_attention_heads
    B = images.shape[0]
    w = sd[p + "embeddings.patch_embedding.weight"]
    x = F.conv2d(images.to(w.dtype), w, bias=None, stride=cfg.patch_size)        # (B, C, g, g)
    x = x.flatten(2).transpose(1, 2)                                                # (B, N, C)
    cls = sd[p + "embeddings.class_embedding"].expand(B, 1, -1)
    x = torch.cat([cls, x], dim=1) + sd[p + "embeddings.position_embedding.weight"][None]
    x = F.layer_norm(x, (C,), sd[p + "pre_layrnorm.weight"], sd[p + "pre_layrnorm.bias"], cfg.layer_norm_eps)
    hs = [x]
    L = cfg.num_hidden_layers if n_layers is None else n_layers
    scale = dh ** -0.5
    for i in range(L):
        q = p + f"encoder.layers.{i}."
        y = F.layer_norm(x, (C,), sd[q + "layer_norm1.weight"], sd[q + "layer_norm1.bias"], cfg.layer_norm_eps)
        T = y.shape[1]
        qh = F.linear(y, sd[q + "self_attn.q_proj.weight"], sd[q + "self_attn.q_proj.bias"]).view(B, T, H, dh).transpose(1, 2)
        kh = F.linear(y, sd[q + "self_attn.k_proj.weight"], sd[q + "self_attn.k_proj.bias"]).view(B, T, H, dh).transpose(1, 2)
        vh = F.linear(y, sd[q + "self_attn.v_proj.weight"], sd[q + "self_attn.v_proj.bias"]).view(B, T, H, dh).transpose(1, 2)
        att = torch.softmax(torch.matmul(qh, kh.transpose(-1, -2)) * scale, dim=-1)
        a = torch.matmul(att, vh).transpose(1, 2).reshape(B, T, C)
        x = x + F.linear(a, sd[q + "self_attn.out_proj.weight"], sd[q + "self_attn.out_proj.bias"])
        y = F.layer_norm(x, (C,), sd[q + "layer_norm2.weight"], sd[q + "layer_norm2.bias"], cfg.layer_norm_eps)
        y = quick_gelu(F.linear(y, sd[q + "mlp.fc1.weight"], sd[q + "mlp.fc1.bias"]))
        x = x + F.linear(y, sd[q + "mlp.fc2.weight"], sd[q + "mlp.fc2.bias"])
        hs.append(x)
    return hs


def layers_needed(cfg: VitConfig, select_layer: int) -> int:
    """hidden_states has L+1 entries; index `select_layer` needs this many encoder layers."""
    idx = select_layer if select_layer >= 0 else cfg.num_hidden_layers + 1 + select_layer
    if not 0 <= idx <= cfg.num_hidden_layers:
        raise IndexError(f"select_layer {select_layer} out of range for {cfg.num_hidden_layers} layers")
    return idx


def tower_forward(sd: Dict[str, Tensor], cfg: VitConfig, images: Tensor, select_layer: int = -2,
                  select_feature: str = "patch") -> Tensor:
    """CLIPVisionTower.forward + feature_select (clip_encoder.py:40-62): hidden_states[select_layer],
    drop token 0 for 'patch', keep for 'cls_patch', ValueError otherwise; cast back to input dtype."""
    n = layers_needed(cfg, select_layer)
    feats = clip_vit_hidden_states(sd, cfg, images, n_layers=n)[n]
    if select_feature == "patch":
        feats = feats[:, 1:]
    elif select_feature != "cls_patch":
        raise ValueError(f"Unexpected select feature: {select_feature}")
    return feats.to(images.dtype)


# ----------------------------------------------------------------------------------------------
# a2 — PositionalEncoding2D (module.py:105-146, utils.py:5-10)
# ----------------------------------------------------------------------------------------------
def pos_encoding_2d(h: int, w: int, C: int, dtype=torch.float32, crop: Optional[int] = None) -> Tensor:
    """(h*w, C) table: channels [0,ch) encode the row index, [ch,2ch) the column index, each as
    interleaved (sin, cos) of pos * inv_freq; cropped to C (module.py:112-145)."""
    ch = int(math.ceil(C / 4) * 2)                                                  # module.py:112
    inv_freq = 1.0 / (10000 ** (torch.arange(0, ch, 2).float() / ch))              # :114
    def emb1d(n):
        s = torch.einsum("i,j->ij", torch.arange(n, dtype=inv_freq.dtype), inv_freq)  # :131-134
        return torch.stack((s.sin(), s.cos()), dim=-1).flatten(-2, -1)             # utils.py:9-10
    emb = torch.zeros((h, w, ch * 2), dtype=dtype)                                  # :137-141
    emb[:, :, :ch] = emb1d(h).unsqueeze(1).to(dtype)                                # :142 (row index)
    emb[:, :, ch:2 * ch] = emb1d(w).to(dtype)                                       # :143 (col index)
    crop = C if crop is None else crop              # :126,145: cropped to the INPUT's channel count (== C for the tokenizer)
    if crop > 2 * ch:
        raise ValueError("input wider than the positional table (the reference's add would fail to broadcast)")
    return emb[:, :, :crop].reshape(h * w, crop)


# ----------------------------------------------------------------------------------------------
# a3 — cluster_dpc_knn (tokenizer.py:78-121)
# ----------------------------------------------------------------------------------------------
def pairwise_dist(x: Tensor) -> Tensor:
    """torch.cdist(x, x) default compute mode (tokenizer.py:82): for more than 25 rows the matmul
    form on augmented vectors [-2x, |x|^2, 1]·[x, 1, |x|^2]^T, clamp_min(0), sqrt (hence a diagonal
    that is not exactly 0 and a matrix that is not bitwise symmetric — SURVEY.md §7 "hard parts");
    for <= 25 rows the direct sqrt(sum (a-b)^2)."""
    N = x.shape[0]
    if N > 25:
        n = x.pow(2).sum(-1, keepdim=True)
        one = torch.ones_like(n)
        a = torch.cat([x.mul(-2), n, one], dim=-1)
        b = torch.cat([x, one, n], dim=-1)
        return a.matmul(b.t()).clamp_min_(0).sqrt_()
    d = x[:, None, :] - x[None, :, :]
    return d.pow(2).sum(-1).sqrt()


@dataclass
class ClusterResult:
    index_down: Tensor      # (L,) int64 — centre token indices, ascending
    idx_cluster: Tensor     # (N,) int64 — cluster id of each token
    score: Tensor           # (1, N)
    density: Tensor         # (N,)
    delta: Tensor           # (N,)
    dist: Tensor            # (N, N) scaled distance matrix (after token_mask, if any)
    fallback: bool          # True when no score exceeded the threshold (topk branch, :104-107)


def cluster_dpc_knn(x: Tensor, k: int, threshold: float, min_cluster_num: int,
                    token_mask: Optional[Tensor] = None, noise: Optional[Tensor] = None) -> ClusterResult:
    """DPC-kNN exactly as tokenizer.py:78-121 composes it, with the density tie-break noise
    (`torch.rand(N) * 1e-6`, :91) as an explicit input (None == zeros)."""
    N, C = x.shape
    dist = pairwise_dist(x) / (C ** 0.5)                                            # :82
    return cluster_from_dist(dist, x.dtype, k, threshold, min_cluster_num, token_mask, noise)


def cluster_from_dist(dist: Tensor, dtype, k: int, threshold: float, min_cluster_num: int,
                      token_mask: Optional[Tensor] = None, noise: Optional[Tensor] = None) -> ClusterResult:
    """tokenizer.py:84-121 given the scaled distance matrix of :82."""
    if token_mask is not None:                                                      # :84-86
        tm = token_mask > 0
        dist = dist * tm[None, :] + (dist.max() + 1) * (~tm[None, :])
    nearest, _ = torch.topk(dist, k=k, dim=-1, largest=False)                       # :88 (self included)
    density = (-(nearest ** 2).mean(dim=-1)).exp()                                  # :90
    if noise is not None:
        density = density + noise.to(density.dtype) * 1e-6                          # :91
    if token_mask is not None:
        density = density * tm                                                      # :93-94
    mask = (density[None, :] > density[:, None]).to(dtype)                          # :96-97  mask[i,j] = rho_j > rho_i
    dist_max = dist.flatten(1).max(dim=-1)[0][None, None]                           # :98  (1,1,N): row-j max, indexed by LAST axis
    delta, _ = (dist * mask + dist_max * (1 - mask)).min(dim=-1)                    # :99  -> (1, N)
    score = delta * density                                                         # :101 -> (1, N)
    index_down = torch.nonzero(score.reshape(-1) > threshold).reshape(-1)           # :103
    fallback = index_down.numel() == 0
    if fallback:                                                                    # :104-107
        _, index_down = torch.topk(score, k=min_cluster_num, dim=-1)
        index_down = torch.sort(index_down).values.reshape(-1)
    idx_cluster = dist[index_down, :].argmin(dim=0)                                 # :111-113 rows = centres
    idx_cluster[index_down] = torch.arange(index_down.numel())                      # :117-119
    return ClusterResult(index_down, idx_cluster, score, density, delta.reshape(-1), dist, fallback)


def cluster_sensitivity(x: Tensor, k: int, threshold: float, min_cluster_num: int,
                        token_mask: Optional[Tensor] = None, noise: Optional[Tensor] = None,
                        trials: int = 16, ulps: float = 4.0, seed: int = 0) -> Dict[str, Tensor]:
    """Which discrete decisions of cluster_dpc_knn survive fp32-rounding-sized perturbations?

    Two correct fp32 implementations of tokenizer.py:82 differ in the summation order of
    |a|^2 + |b|^2 - 2 a.b, i.e. by a few ulps of (|a|^2 + |b|^2) in d^2 — a LARGE relative error for
    near-duplicate tokens, which is why the reference itself adds tie-break noise (:91); densities that
    tie within rounding flip the `rho_j > rho_i` mask and change delta discretely.  This runs the
    algorithm in fp64 on the exact d^2 and on `trials` copies perturbed by N(0,1) * ulps * 2^-24 *
    (|a|^2 + |b|^2); a decision is *certain* when all runs agree.  Parity tests demand bit-exact
    integers for certain decisions and self-consistency for the rest (check_cluster_parity)."""
    xd = x.double()
    N, C = xd.shape
    n = xd.pow(2).sum(-1)
    d2 = (n[:, None] + n[None, :] - 2 * xd @ xd.t()).clamp_min(0)
    scale = ulps * 2.0 ** -24 * (n[:, None] + n[None, :])
    g = torch.Generator().manual_seed(seed)
    nz = None if noise is None else noise.double()
    runs = []
    for t in range(trials + 1):
        p = d2 if t == 0 else (d2 + scale * torch.randn(N, N, generator=g, dtype=torch.float64)).clamp_min(0)
        runs.append(cluster_from_dist((p / C).sqrt(), torch.float64, k, threshold, min_cluster_num, token_mask, nz))
    sel = torch.stack([torch.zeros(N, dtype=torch.bool).index_fill_(0, r.index_down, True) for r in runs])
    centre_in, centre_out = sel.all(0), ~sel.any(0)
    centres_certain = bool((centre_in | centre_out).all())
    labels = torch.stack([r.idx_cluster for r in runs])
    assign_certain = (labels == labels[0:1]).all(0) if centres_certain else torch.zeros(N, dtype=torch.bool)
    scores = torch.stack([r.score.reshape(-1) for r in runs])
    return dict(centre_in=centre_in, centre_out=centre_out, centres_certain=centres_certain,
                assign_certain=assign_certain, index_down=runs[0].index_down, idx_cluster=runs[0].idx_cluster,
                score=scores[0], score_lo=scores.min(0).values, score_hi=scores.max(0).values)


def check_score(got_score: Tensor, sens: Dict[str, Tensor], rtol: float = 1e-3) -> None:
    """`score` is a float by-product (the integers are the contract): it must lie inside the envelope
    the perturbed fp64 runs span (a density near-tie moves delta, hence score, discretely), widened by
    `rtol` for the d^2 cancellation error of near-duplicate tokens."""
    g = got_score.reshape(-1).double()
    lo, hi = sens["score_lo"] * (1 - rtol) - 1e-7, sens["score_hi"] * (1 + rtol) + 1e-7
    bad = (g < lo) | (g > hi)
    assert not bool(bad.any()), f"{int(bad.sum())} scores outside the fp32-perturbation envelope, e.g. token {int(bad.nonzero()[0])}"


def check_cluster_parity(got_index_down: Tensor, got_idx_cluster: Tensor, ref_index_down: Tensor,
                         ref_idx_cluster: Tensor, iv: Dict[str, Tensor]) -> Dict[str, int]:
    """The integer contract: wherever cluster_sensitivity says a decision is certain, `got` must
    equal `ref` bit for bit; everywhere it must be self-consistent (sorted centres, centres own
    themselves, certainly-in tokens selected, certainly-out tokens not).  Raises AssertionError."""
    N = got_idx_cluster.numel()
    L = got_index_down.numel()
    assert torch.equal(got_index_down, torch.sort(got_index_down).values) and got_index_down.unique().numel() == L
    assert int(got_idx_cluster.min()) >= 0 and int(got_idx_cluster.max()) < L
    assert torch.equal(got_idx_cluster[got_index_down], torch.arange(L)), "a centre does not own itself"
    sel = torch.zeros(N, dtype=torch.bool); sel[got_index_down] = True
    assert bool(sel[iv["centre_in"]].all()), "a certainly-selected centre is missing"
    assert not bool(sel[iv["centre_out"]].any()), "a certainly-rejected token was selected"
    stats = dict(centres_certain=int(iv["centres_certain"]), tokens_certain=0, tokens_compared=0)
    if iv["centres_certain"]:
        assert torch.equal(got_index_down, ref_index_down), "centre set differs although every centre decision is certain"
        ok = iv["assign_certain"]
        stats["tokens_certain"] = int(ok.sum())
        bad = (got_idx_cluster != ref_idx_cluster) & ok
        assert not bool(bad.any()), f"{int(bad.sum())} tokens with a certain assignment differ"
    if torch.equal(got_index_down, ref_index_down):
        stats["tokens_compared"] = N
        stats["tokens_equal"] = int((got_idx_cluster == ref_idx_cluster).sum())
    return stats


# ----------------------------------------------------------------------------------------------
# a5 — Block / Attention / Mlp (module.py:29-100)
# ----------------------------------------------------------------------------------------------
def block_forward(sd: Dict[str, Tensor], prefix: str, x: Tensor, nheads: int, depth: int,
                  eps: float = 1e-5) -> Tensor:
    """`Block.forward` (module.py:95-100) on x of shape (n, C), eval mode (dropouts are identity):
    `depth` attention sub-layers sharing ONE norm1 (:81,88), then ONE norm2 + Mlp (:98).
    Attention: fused qkv Linear with bias (:56,63), scale d_h^-0.5 (:54), softmax (:67), proj (:58).
    Mlp: fc1, exact-erf GELU (nn.GELU default), fc2 (:39-45)."""
    n, C = x.shape
    dh = C // nheads
    g1, b1 = sd[prefix + "norm1.weight"], sd[prefix + "norm1.bias"]
    for i in range(depth):
        a = prefix + f"layers.{i}.1."
        y = F.layer_norm(x, (C,), g1, b1, eps)
        qkv = F.linear(y, sd[a + "qkv.weight"], sd[a + "qkv.bias"]).reshape(n, 3, nheads, dh).permute(1, 2, 0, 3)
        q, kk, v = qkv[0], qkv[1], qkv[2]                                           # (H, n, dh)
        att = torch.softmax((q @ kk.transpose(-2, -1)) * (dh ** -0.5), dim=-1)
        o = (att @ v).transpose(0, 1).reshape(n, C)
        x = x + F.linear(o, sd[a + "proj.weight"], sd[a + "proj.bias"])
    y = F.layer_norm(x, (C,), sd[prefix + "norm2.weight"], sd[prefix + "norm2.bias"], eps)
    y = F.gelu(F.linear(y, sd[prefix + "mlp.fc1.weight"], sd[prefix + "mlp.fc1.bias"]))
    return x + F.linear(y, sd[prefix + "mlp.fc2.weight"], sd[prefix + "mlp.fc2.bias"])


# ----------------------------------------------------------------------------------------------
# a4 — group_encoding (tokenizer.py:123-155)
# ----------------------------------------------------------------------------------------------
def group_encoding(sd: Dict[str, Tensor], hc: HeadConfig, x: Tensor, labels: Tensor) -> Tensor:
    """For each unique label (ascending, :141): inner_encoder on that cluster's member tokens alone
    (:150), then the uniform mean over members (:151); stack (:153).  `centers` is unused (:123)."""
    out = []
    for lab in labels.unique():
        m = labels == lab
        out.append(block_forward(sd, "inner_encoder.", x[m], hc.nheads, hc.inner_cluster_layers).mean(dim=0))
    return torch.stack(out, dim=0)


# ----------------------------------------------------------------------------------------------
# a7 — SetokTokenizer.forward per image (tokenizer.py:157-182, repairs D1/D2 of SURVEY.md §0.2)
# ----------------------------------------------------------------------------------------------
@dataclass
class HeadResult:
    tokens: Tensor          # (L, token_feat_dim)
    idx_cluster: Tensor     # (N,) int64
    score: Tensor           # (1, N)
    index_down: Tensor      # (L,) int64
    x: Tensor               # (N, C) features + positional encoding
    group: Tensor           # (L, C) after group_encoding
    inter: Tensor           # (L, C) after inter_encoder


def head_forward(sd: Dict[str, Tensor], hc: HeadConfig, feats: Tensor, k=None, threshold=None,
                 token_mask: Optional[Tensor] = None, noise: Optional[Tensor] = None) -> HeadResult:
    N, C = feats.shape
    h = w = int(math.sqrt(N))                                                       # :164
    x = feats + pos_encoding_2d(h, w, C, feats.dtype)                               # :165-168
    _threshold = threshold if threshold else hc.threshold                          # :171 (truthiness)
    _k = k if k else hc.min_cluster_num                                             # :172
    cr = cluster_dpc_knn(x, _k, _threshold, hc.min_cluster_num, token_mask, noise)  # :174
    group = group_encoding(sd, hc, x, cr.idx_cluster)                               # :177-178
    inter = block_forward(sd, "inter_encoder.", group, hc.nheads, hc.intra_cluster_layers)  # :179 (+D2)
    tokens = F.linear(inter, sd["out.weight"], sd["out.bias"])                      # :180
    return HeadResult(tokens, cr.idx_cluster, cr.score, cr.index_down, x, group, inter)


def encode(sd: Dict[str, Tensor], vc: VitConfig, hc: HeadConfig, images: Tensor, k=None, threshold=None,
           noise: Optional[Tensor] = None) -> Tuple[Tensor, List[HeadResult]]:
    """Tower on the batch, then the per-image head (the batch dimension is a loop — D1)."""
    sd = normalise_tower_keys(sd)
    feats = tower_forward(sd, vc, images, hc.mm_vision_select_layer, hc.mm_vision_select_feature)
    res = [head_forward(sd, hc, feats[i], k, threshold, None, None if noise is None else noise[i])
           for i in range(feats.shape[0])]
    return feats, res


# ----------------------------------------------------------------------------------------------
# §8(f) row 4 — gradients of the trainable head (inner_encoder, inter_encoder, out) for a training step.
# The clustering runs under no_grad in the reference (tokenizer.py:79) and the tower is frozen (clip_encoder.py:50,
# unfreeze_mm_vision_tower=False), so the backward pass covers a4-a6 only and stops at the head's parameters.
# ----------------------------------------------------------------------------------------------
HEAD_PARAM_PREFIXES = ("inner_encoder.", "inter_encoder.", "out.")


def head_param_grads(sd: Dict[str, Tensor], hc: HeadConfig, feats: Sequence[Tensor], upstream: Sequence[Tensor], k=None,
                     threshold=None, noise: Optional[Sequence[Tensor]] = None) -> Tuple[Dict[str, Tensor], List[HeadResult]]:
    """d/d(theta) of  L = sum_i <tokens_i, upstream_i>  by autograd through the oracle's own forward (`head_forward`), i.e. the
    parameter gradients a training step sees when dL/dtokens_i = upstream_i arrives from the projector / LLM.  feats[i]: (N, C) tower
    features of image i; upstream[i]: (L_i, token_feat_dim)."""
    params = {n: (v.detach().clone().requires_grad_(True) if n.startswith(HEAD_PARAM_PREFIXES) and ".0." not in n.split("layers.")[-1][:4]
                  else v.detach()) for n, v in sd.items()}
    for n in list(params):                                   # `layers.{i}.0.*` are aliases of norm1 (module.py:87-88): one shared parameter
        if ".layers." in n and n.split(".layers.")[1].split(".")[1] == "0":
            params[n] = params[n.split(".layers.")[0] + ".norm1." + n.split(".")[-1]]
    loss = 0.0
    res = []
    for i, f in enumerate(feats):
        r = head_forward(params, hc, f, k, threshold, None, None if noise is None else noise[i])
        assert tuple(r.tokens.shape) == tuple(upstream[i].shape), (r.tokens.shape, upstream[i].shape)
        loss = loss + (r.tokens * upstream[i]).sum()
        res.append(r)
    loss.backward()
    grads = {n: p.grad for n, p in params.items() if p.requires_grad and p.grad is not None
             and not (".layers." in n and n.split(".layers.")[1].split(".")[1] == "0")}      # aliases of norm1 reported once
    return grads, res


# ----------------------------------------------------------------------------------------------
# a8 — mm_in_projector (src/model/multimodal_projector/builder.py:33-64) + encode_images
#      (src/model/setokim_arch.py:206-211)
# ----------------------------------------------------------------------------------------------
def projector_forward(psd: Dict[str, Tensor], projector_type: str, x: Tensor) -> Tensor:
    """'linear' | 'mlp{N}x_gelu[_Norm]' | 'identity'.  Keys follow nn.Sequential numbering."""
    import re
    if projector_type == "identity":
        return x
    if projector_type == "linear":
        return F.linear(x, psd["weight"], psd["bias"])
    use_norm = "_Norm" in projector_type
    m = re.match(r"^mlp(\d+)x_gelu$", projector_type.replace("_Norm", ""))
    if not m:
        raise ValueError(f"Unknown projector type: {projector_type}")
    depth = int(m.group(1))
    i = 0
    x = F.linear(x, psd[f"{i}.weight"], psd[f"{i}.bias"]); i += 1
    if use_norm:
        x = F.layer_norm(x, (x.shape[-1],), psd[f"{i}.weight"], psd[f"{i}.bias"], 1e-5); i += 1
    for _ in range(1, depth):
        x = F.gelu(x); i += 1
        x = F.linear(x, psd[f"{i}.weight"], psd[f"{i}.bias"]); i += 1
    return x


def encode_images(sd, psd, projector_type, vc: VitConfig, hc: HeadConfig, images: Tensor, k=None,
                  threshold=None, noise=None) -> List[Tensor]:
    """encode_images (setokim_arch.py:206-211) with the ragged result D3 requires: per image
    (L_i, D) after mm_in_projector."""
    _, res = encode(sd, vc, hc, images, k, threshold, noise)
    return [projector_forward(psd, projector_type, r.tokens) for r in res]


# ----------------------------------------------------------------------------------------------
# a9 — reconstruction decoder (cfg 3): SetokDeTokenizer (detokenizer.py:14-120)
# ----------------------------------------------------------------------------------------------
@dataclass
class DetokConfig:
    """Constructor arguments of SetokDeTokenizer (detokenizer.py:15-30) plus the BertConfig fields the
    Q-Former arithmetic reads (bert-base-uncased values: detokenizer.py:80; they are BertConfig()'s
    defaults).  `hidden_dim` must equal `mapper_hidden`: the queries (1, Q, hidden_dim) go straight
    into BertEmbeddings.LayerNorm(hidden_size) (module.py:163,203) — train_setokim.py:361 sets 768."""
    token_feat_dim: int = 4096
    hidden_dim: int = 768
    patch_size: int = 14
    image_size: int = 256
    decoder_embed_dim: int = 768
    decoder_nheads: int = 16
    decoder_depth: int = 16
    mlp_ratio: float = 4.0
    num_hidden_layers: int = 6
    cross_attention_freq: int = 2
    mapper_hidden: int = 768
    mapper_heads: int = 12
    mapper_intermediate: int = 3072
    mapper_eps: float = 1e-12
    norm_eps: float = 1e-5          # norm_layer = nn.LayerNorm (detokenizer.py:25) -> default eps

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size                                   # detokenizer.py:36

    @property
    def num_queries(self) -> int:
        return self.grid * self.grid                                                # :37


def bert_attention(sd: Dict[str, Tensor], p: str, x: Tensor, kv: Tensor, add_mask: Optional[Tensor],
                   heads: int, eps: float) -> Tensor:
    """`BertAttention.forward` (module.py:418-443) in eval mode on x (B, Q, H): BertSelfAttention
    (:267-373: query from x, key/value from `kv` — x itself for self-attention, the encoder states
    for cross-attention :283-286 —, scores / sqrt(d_h) :342, + additive mask :343-345, softmax :348,
    context :360-364) then BertSelfOutput (:383-387: dense, LayerNorm(dense + input))."""
    B, Q, H = x.shape
    dh = H // heads
    def split(t):                                                                   # transpose_for_scores :259-265
        return t.reshape(t.shape[0], t.shape[1], heads, dh).permute(0, 2, 1, 3)
    q = split(F.linear(x, sd[p + "self.query.weight"], sd[p + "self.query.bias"]))
    k = split(F.linear(kv, sd[p + "self.key.weight"], sd[p + "self.key.bias"]))
    v = split(F.linear(kv, sd[p + "self.value.weight"], sd[p + "self.value.bias"]))
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh)
    if add_mask is not None:
        s = s + add_mask
    ctx = torch.matmul(torch.softmax(s, dim=-1), v).permute(0, 2, 1, 3).reshape(B, Q, H)
    y = F.linear(ctx, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"])
    return F.layer_norm(y + x, (H,), sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], eps)


def qformer_forward(sd: Dict[str, Tensor], dc: DetokConfig, query_embeds: Tensor, enc: Tensor,
                    enc_mask: Optional[Tensor], prefix: str = "mapper.") -> Tensor:
    """`BertModel.forward` (module.py:852-1014) as SetokDeTokenizer calls it (detokenizer.py:105-109:
    query_embeds only, no input_ids, not a decoder).  Embeddings = LayerNorm(query_embeds) (:200-204);
    the self-attention mask is all ones -> additive 0 (:923-939,849); the encoder mask m becomes
    (1 - m) * -10000 broadcast over heads and queries (invert_attention_mask, :962-973).
    Per layer (BertLayer.forward :500-572, query_length == Q): self-attention; cross-attention when
    layer_num % cross_attention_freq == 0 (:484-491,532-546); then the QUERY feed-forward
    (intermediate_query: dense + erf-GELU :450-453; output_query: dense, LayerNorm(. + input) :464-468)."""
    H = dc.mapper_hidden
    eps = dc.mapper_eps
    x = F.layer_norm(query_embeds, (H,), sd[prefix + "embeddings.LayerNorm.weight"],
                     sd[prefix + "embeddings.LayerNorm.bias"], eps)
    add = None
    if enc_mask is not None:
        add = ((1.0 - enc_mask.to(x.dtype)) * -10000.0)[:, None, None, :]
    zero = torch.zeros((x.shape[0], 1, 1, x.shape[1]), dtype=x.dtype)            # (1 - 1) * -10000
    for i in range(dc.num_hidden_layers):
        lp = prefix + f"encoder.layer.{i}."
        x = bert_attention(sd, lp + "attention.", x, x, zero, dc.mapper_heads, eps)
        if i % dc.cross_attention_freq == 0:
            x = bert_attention(sd, lp + "crossattention.", x, enc, add, dc.mapper_heads, eps)
        y = F.gelu(F.linear(x, sd[lp + "intermediate_query.dense.weight"], sd[lp + "intermediate_query.dense.bias"]))
        y = F.linear(y, sd[lp + "output_query.dense.weight"], sd[lp + "output_query.dense.bias"])
        x = F.layer_norm(y + x, (H,), sd[lp + "output_query.LayerNorm.weight"], sd[lp + "output_query.LayerNorm.bias"], eps)
    return x


def vit_block_forward(sd: Dict[str, Tensor], p: str, x: Tensor, heads: int, eps: float) -> Tensor:
    """timm==0.9.16 `vision_transformer.Block.forward` (third party, pyproject.toml:22, NOT installed here —
    restated from its published algorithm; call site detokenizer.py:49-51 with qkv_bias=True, no
    qk_norm, no LayerScale, drop_path 0): x + proj(attn(norm1 x)); x + fc2(gelu(fc1(norm2 x))).
    Attention: fused qkv Linear -> (3, heads, d_h), scale d_h^-0.5, softmax, proj."""
    B, T, C = x.shape
    dh = C // heads
    y = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
    qkv = F.linear(y, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]).reshape(B, T, 3, heads, dh).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = torch.softmax((q * dh ** -0.5) @ k.transpose(-2, -1), dim=-1)
    o = (att @ v).transpose(1, 2).reshape(B, T, C)
    x = x + F.linear(o, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
    y = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps)
    y = F.gelu(F.linear(y, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
    return x + F.linear(y, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])


def detokenizer_forward(sd: Dict[str, Tensor], dc: DetokConfig, x: Tensor, attention_masks: Optional[Tensor],
                        return_stages: bool = False):
    """`SetokDeTokenizer.forward` (detokenizer.py:101-120) on padded tokens x (B, L, token_feat_dim) and
    mask (B, L).  The reference computes this value and then returns None (defect D5); the restatement
    returns it: (B, Q, decoder_embed_dim)."""
    B = x.shape[0]
    mask_tokens = sd["mask_tokens"].expand(B, -1, -1)                               # :103
    enc = F.linear(x, sd["mapper_fc_in.weight"], sd["mapper_fc_in.bias"])          # :104
    mapped = qformer_forward(sd, dc, mask_tokens, enc, attention_masks)             # :105-109
    y = F.linear(mapped, sd["decoder_fc_in.weight"], sd["decoder_fc_in.bias"])     # :111
    pos = pos_encoding_2d(dc.grid, dc.grid, dc.hidden_dim, y.dtype, crop=dc.decoder_embed_dim)   # :52,112-114 (module.py:145)
    y = y + pos[None]                                                               # :115
    z = y
    for i in range(dc.decoder_depth):                                               # :117-118
        z = vit_block_forward(sd, f"pixel_decoder.{i}.", z, dc.decoder_nheads, dc.norm_eps)
    out = F.layer_norm(z, (dc.decoder_embed_dim,), sd["decoder_norm.weight"], sd["decoder_norm.bias"], dc.norm_eps)  # :120
    if return_stages:
        return dict(enc=enc, mapped=mapped, dec_in=y, out=out)
    return out


def init_detok_weights(dc: DetokConfig, seed: int = 3, dtype=torch.float32) -> Dict[str, Tensor]:
    """Seeded synthetic weights under the reference's state-dict names: xavier-uniform Linear weights /
    unit LayerNorms for the modules `_init_weights` touches (detokenizer.py:57-69), N(0, 0.02) queries
    (:39-41) and Q-Former Linears (BertPreTrainedModel._init_weights, initializer_range 0.02).  Biases and
    LayerNorm affine parameters get small random values so that every term is exercised."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, Tensor] = {}
    def lin(name, o, i, std=None):
        if std is None:
            a = math.sqrt(6.0 / (i + o))
            sd[name + ".weight"] = (torch.rand(o, i, generator=g) * 2 - 1) * a
        else:
            sd[name + ".weight"] = torch.randn(o, i, generator=g) * std
        sd[name + ".bias"] = torch.randn(o, generator=g) * 0.02
    def ln(name, n):
        sd[name + ".weight"] = 1.0 + 0.05 * torch.randn(n, generator=g)
        sd[name + ".bias"] = 0.02 * torch.randn(n, generator=g)
    H, D = dc.mapper_hidden, dc.decoder_embed_dim
    sd["mask_tokens"] = torch.randn(1, dc.num_queries, dc.hidden_dim, generator=g) * 0.02
    lin("mapper_fc_in", dc.hidden_dim, dc.token_feat_dim)
    lin("decoder_fc_in", D, dc.hidden_dim)
    ln("decoder_norm", D)
    ln("mapper.embeddings.LayerNorm", H)
    for i in range(dc.num_hidden_layers):
        lp = f"mapper.encoder.layer.{i}."
        atts = ["attention."] + (["crossattention."] if i % dc.cross_attention_freq == 0 else [])
        for a in atts:
            kin = dc.hidden_dim if a == "crossattention." else H                    # encoder_width = hidden_dim (detokenizer.py:53,82)
            lin(lp + a + "self.query", H, H, 0.02)
            lin(lp + a + "self.key", H, kin, 0.02)
            lin(lp + a + "self.value", H, kin, 0.02)
            lin(lp + a + "output.dense", H, H, 0.02)
            ln(lp + a + "output.LayerNorm", H)
        lin(lp + "intermediate_query.dense", dc.mapper_intermediate, H, 0.02)
        lin(lp + "output_query.dense", H, dc.mapper_intermediate, 0.02)
        ln(lp + "output_query.LayerNorm", H)
    ff = int(D * dc.mlp_ratio)
    for i in range(dc.decoder_depth):
        p = f"pixel_decoder.{i}."
        ln(p + "norm1", D); lin(p + "attn.qkv", 3 * D, D); lin(p + "attn.proj", D, D)
        ln(p + "norm2", D); lin(p + "mlp.fc1", ff, D); lin(p + "mlp.fc2", D, ff)
    return {k: v.to(dtype) for k, v in sd.items()}


# ----------------------------------------------------------------------------------------------
# §8(f) row 1 — prepare_inputs_labels_for_multimodal (setokim_arch.py:213-355): the ragged splice
# ----------------------------------------------------------------------------------------------
IGNORE_INDEX, IMAGE_TOKEN_INDEX, TARGET_TOKEN_INDEX = -100, -200, -300          # src/constants.py:7-8,15


def splice_multimodal(input_ids: Tensor, position_ids: Optional[Tensor], attention_mask: Optional[Tensor],
                      labels: Optional[Tensor], image_features: Sequence[Tensor], embed_weight: Tensor,
                      max_length: Optional[int] = None, padding_side: str = "right"):
    """The data path of `prepare_inputs_labels_for_multimodal` after `encode_images` (setokim_arch.py:241-353), given the
    per-image token matrices `image_features[i]` (L_i, D) and the LLM's embedding table.  Returns
    (position_ids, attention_mask, inputs_embeds, labels) with the reference's None conventions (:341-353).

    Per sequence (:262-308): padding removed by the mask (:258-259); every IMAGE_TOKEN_INDEX placeholder is replaced by the
    next image's tokens (labels IGNORE_INDEX, :292-293), text tokens by their embedding rows; a sequence WITHOUT a placeholder
    still consumes one image index (:264-271).  Then truncation (:311-314), padding to the batch maximum on the configured
    side with zero rows / IGNORE_INDEX / False / 0 (:317-337), TARGET_TOKEN_INDEX labels -> IGNORE_INDEX (:344)."""
    B, T = input_ids.shape
    am = torch.ones_like(input_ids, dtype=torch.bool) if attention_mask is None else attention_mask.bool()
    lab = torch.full_like(input_ids, IGNORE_INDEX) if labels is None else labels
    D = embed_weight.shape[1]
    rows_all, labs_all = [], []
    img = 0
    for b in range(B):
        ids, lb = input_ids[b][am[b]], lab[b][am[b]]
        rows, labs = [], []
        n_img = int((ids == IMAGE_TOKEN_INDEX).sum())
        if n_img == 0:
            _ = image_features[img]                                         # :265 (indexing happens, so it can raise)
            img += 1
        for t in range(ids.shape[0]):
            if int(ids[t]) == IMAGE_TOKEN_INDEX:
                f = image_features[img]; img += 1
                rows.append(f)
                labs.append(torch.full((f.shape[0],), IGNORE_INDEX, dtype=lb.dtype))
            else:
                rows.append(embed_weight[int(ids[t])][None])
                labs.append(lb[t:t + 1])
        r = torch.cat(rows, 0) if rows else embed_weight.new_zeros((0, D))
        l = torch.cat(labs, 0) if labs else lb.new_zeros((0,))
        if max_length is not None:
            r, l = r[:max_length], l[:max_length]
        rows_all.append(r); labs_all.append(l)
    max_len = max(r.shape[0] for r in rows_all)
    emb = embed_weight.new_zeros((B, max_len, D))
    new_labels = torch.full((B, max_len), IGNORE_INDEX, dtype=lab.dtype)
    new_am = torch.zeros((B, max_len), dtype=torch.bool)
    new_pos = torch.zeros((B, max_len), dtype=torch.long if position_ids is None else position_ids.dtype)
    for b, (r, l) in enumerate(zip(rows_all, labs_all)):
        n = r.shape[0]
        if n == 0:
            continue
        sl = slice(max_len - n, max_len) if padding_side == "left" else slice(0, n)
        emb[b, sl] = r; new_labels[b, sl] = l; new_am[b, sl] = True
        new_pos[b, sl] = torch.arange(n, dtype=new_pos.dtype)
    new_labels[new_labels == TARGET_TOKEN_INDEX] = IGNORE_INDEX
    return (None if position_ids is None else new_pos,
            None if attention_mask is None else new_am.to(attention_mask.dtype),
            emb,
            None if labels is None else new_labels)


def splice_inputs(seed, B, T, V, D, max_imgs=3, pad=True):
    """Seeded inputs of prepare_inputs_labels_for_multimodal: ids with placeholders (-200), a prefix mask, labels with a few
    TARGET (-300) entries, ragged image tokens.  Regenerates bit-exactly from the seed (torch CPU generator)."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, V, (B, T), generator=g)
    am = torch.ones(B, T, dtype=torch.long)
    n_need = 0
    for b in range(B):
        n_valid = int(torch.randint(1, T + 1, (1,), generator=g)) if pad else T
        am[b, n_valid:] = 0
        n_img = int(torch.randint(0, max_imgs + 1, (1,), generator=g))
        n_img = min(n_img, n_valid)
        where = torch.randperm(n_valid, generator=g)[:n_img]
        ids[b, where] = IMAGE_TOKEN_INDEX
        n_need += max(n_img, 1)
    labels = torch.where(am.bool(), torch.randint(0, V, (B, T), generator=g), torch.full((B, T), IGNORE_INDEX))
    tgt = torch.rand(B, T, generator=g) < 0.05
    labels[tgt & am.bool()] = TARGET_TOKEN_INDEX
    feats = [torch.randn(int(torch.randint(1, 9, (1,), generator=g)), D, generator=g) for _ in range(n_need)]
    W = torch.randn(V, D, generator=g)
    return ids, am, labels, feats, W


# ----------------------------------------------------------------------------------------------
# §8(f) row 2 — the pixel head the reference never defines.  SetokDeTokenizer.forward returns None after decoder_norm
# (detokenizer.py:101-120) while SeTok.forward passes the result to a pixel-space loss as an image (model.py:75-76,91): the build's
# definition is `to_pixels` (Linear decoder_embed_dim -> patch^2 * 3 per query) + the rearrangement below + the reference's own pixel terms.
# ----------------------------------------------------------------------------------------------
def unpatchify(patches: Tensor, B: int, gh: int, gw: int, p: int) -> Tensor:
    """(B*gh*gw, 3 p^2) rows with columns (pi, qi, c), c fastest -> (B, 3, gh*p, gw*p): 'n (h w) (p q c) -> n c (h p) (w q)'."""
    x = patches.reshape(B, gh, gw, p, p, 3)
    return torch.einsum("nhwpqc->nchpwq", x).reshape(B, 3, gh * p, gw * p)


def pixel_loss(pred: Tensor, target: Tensor, kind: str = "mse") -> Tensor:
    """"mse": WeightedMSELoss.forward without a mask (src/model/loss/mse.py:9-19, weight 1): nn.MSELoss(reduction='none'), mean over
    (C, H, W), mean over the batch.  "l1": the pixel term of the GAN loss, torch.abs(inputs - reconstructions) then torch.mean
    (src/model/loss/discriminator.py:161,170)."""
    if kind == "mse":
        return ((pred - target) ** 2).mean([-3, -2, -1]).mean()
    return torch.abs(target - pred).mean()


def stage2_downstream(embeds, new_labels, w_down):
    """The small stand-in for the LLM behind inputs_embeds in the stage-2 gradient tests (test infrastructure, shared by the golden generator —
    where it sits behind the REFERENCE's projector and splice — and the GPU test): a tanh, a vocabulary projection, the shifted cross entropy of
    setokim_llama.py:145-160's shape and a small log-partition term that gives EVERY position (image rows included, whose next-token labels
    are mostly IGNORE_INDEX) a gradient."""
    logits = torch.tanh(embeds.float()) @ w_down.float().t()
    V = logits.shape[-1]
    ce = torch.nn.functional.cross_entropy(logits[:, :-1].reshape(-1, V), new_labels[:, 1:].reshape(-1), ignore_index=-100)
    return ce + 0.05 * torch.logsumexp(logits, dim=-1).mean()


# ----------------------------------------------------------------------------------------------
# §8(f) last row — the LLM prefill of cfg 5: SetokimLlamaForCausalLM.forward (setokim_llama.py:94-143) = the splice above, then
# `self.model(inputs_embeds=..., attention_mask=..., position_ids=...)` and `self.lm_head` (:130-143).  `self.model` is HuggingFace
# `transformers` LlamaModel (third party; reference pin transformers==4.46.3, pyproject.toml:18; 5.15.0 installed here — same
# arithmetic on the eager attention path): restated here from its published algorithm and pinned against the installed implementation.
# ----------------------------------------------------------------------------------------------
@dataclass
class LlamaConfigLite:
    """Subset of HF LlamaConfig the prefill arithmetic reads (defaults = Vicuna-7B v1.5 / Llama-2-7B)."""
    hidden_size: int = 4096
    intermediate_size: int = 11008
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    num_key_value_heads: int = 32
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    vocab_size: int = 32000

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


def llama_rmsnorm(x: Tensor, w: Tensor, eps: float) -> Tensor:
    """LlamaRMSNorm.forward: statistics in fp32, the normalised value cast back to the input dtype BEFORE the weight multiply."""
    h = x.float()
    h = h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + eps)
    return w * h.to(x.dtype)


def llama_rope_tables(position_ids: Tensor, head_dim: int, theta: float, dtype) -> Tuple[Tensor, Tensor]:
    """LlamaRotaryEmbedding.forward (default rope): fp32 inv_freq x position, emb = cat(freqs, freqs), cos / sin cast to dtype."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    freqs = position_ids[:, :, None].float() * inv_freq[None, None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def _rotate_half(x: Tensor) -> Tensor:
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def llama_forward(sd: Dict[str, Tensor], lc: LlamaConfigLite, inputs_embeds: Tensor, attention_mask: Optional[Tensor],
                  position_ids: Optional[Tensor] = None, with_logits: bool = True, prefix: str = "model."):
    """LlamaModel.forward on inputs_embeds (B, T, D) (eager attention, no cache) + lm_head (setokim_llama.py:130-143).
    attention_mask (B, T): 1 = token; the additive mask is causal AND key-padding (masked = dtype min, as HF builds it).
    Returns (hidden_states after the final norm, logits or None)."""
    B, T, D = inputs_embeds.shape
    H, Hkv, dh = lc.num_attention_heads, lc.num_key_value_heads, lc.head_dim
    if position_ids is None:
        position_ids = torch.arange(T)[None].expand(B, T)
    cos, sin = llama_rope_tables(position_ids, dh, lc.rope_theta, inputs_embeds.dtype)
    cos, sin = cos[:, None], sin[:, None]
    neg = torch.finfo(inputs_embeds.dtype).min
    allow = torch.tril(torch.ones(T, T, dtype=torch.bool))[None, None]
    if attention_mask is not None:
        allow = allow & attention_mask.bool()[:, None, None, :]
    add = torch.zeros((B, 1, T, T), dtype=inputs_embeds.dtype).masked_fill(~allow, neg)
    x = inputs_embeds
    for i in range(lc.num_hidden_layers):
        p = prefix + f"layers.{i}."
        y = llama_rmsnorm(x, sd[p + "input_layernorm.weight"], lc.rms_norm_eps)
        q = F.linear(y, sd[p + "self_attn.q_proj.weight"]).view(B, T, H, dh).transpose(1, 2)
        k = F.linear(y, sd[p + "self_attn.k_proj.weight"]).view(B, T, Hkv, dh).transpose(1, 2)
        v = F.linear(y, sd[p + "self_attn.v_proj.weight"]).view(B, T, Hkv, dh).transpose(1, 2)
        q = (q * cos) + (_rotate_half(q) * sin)
        k = (k * cos) + (_rotate_half(k) * sin)
        if Hkv != H:
            k = k.repeat_interleave(H // Hkv, dim=1); v = v.repeat_interleave(H // Hkv, dim=1)
        w = torch.matmul(q, k.transpose(2, 3)) * dh ** -0.5 + add
        w = torch.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
        o = torch.matmul(w, v).transpose(1, 2).reshape(B, T, D)
        x = x + F.linear(o, sd[p + "self_attn.o_proj.weight"])
        y = llama_rmsnorm(x, sd[p + "post_attention_layernorm.weight"], lc.rms_norm_eps)
        g = F.silu(F.linear(y, sd[p + "mlp.gate_proj.weight"])) * F.linear(y, sd[p + "mlp.up_proj.weight"])
        x = x + F.linear(g, sd[p + "mlp.down_proj.weight"])
    hidden = llama_rmsnorm(x, sd[prefix + "norm.weight"], lc.rms_norm_eps)
    logits = F.linear(hidden, sd["lm_head.weight"]) if with_logits else None
    return hidden, logits


def init_llama_weights(lc: LlamaConfigLite, seed: int = 7, dtype=torch.float32) -> Dict[str, Tensor]:
    """Seeded synthetic weights under HF LlamaForCausalLM's state-dict names (N(0, 0.02) Linears / embeddings as HF initialises,
    RMSNorm weights near 1)."""
    g = torch.Generator().manual_seed(seed)
    D, Fd = lc.hidden_size, lc.intermediate_size
    kv = lc.num_key_value_heads * lc.head_dim
    sd = {"model.embed_tokens.weight": torch.randn(lc.vocab_size, D, generator=g) * 0.02,
          "lm_head.weight": torch.randn(lc.vocab_size, D, generator=g) * 0.02,
          "model.norm.weight": 1.0 + 0.05 * torch.randn(D, generator=g)}
    for i in range(lc.num_hidden_layers):
        p = f"model.layers.{i}."
        sd[p + "input_layernorm.weight"] = 1.0 + 0.05 * torch.randn(D, generator=g)
        sd[p + "post_attention_layernorm.weight"] = 1.0 + 0.05 * torch.randn(D, generator=g)
        for n, (o, ii) in {"self_attn.q_proj": (D, D), "self_attn.k_proj": (kv, D), "self_attn.v_proj": (kv, D), "self_attn.o_proj": (D, D),
                           "mlp.gate_proj": (Fd, D), "mlp.up_proj": (Fd, D), "mlp.down_proj": (D, Fd)}.items():
            sd[p + n + ".weight"] = torch.randn(o, ii, generator=g) * 0.02 * (2.0 if "proj" in n else 1.0)
    return {k: v.to(dtype) for k, v in sd.items()}


def llama_inputs(lc, seed, B, T, padding):
    g = torch.Generator().manual_seed(1000 + seed)
    x = torch.randn(B, T, lc.hidden_size, generator=g)
    am = torch.ones(B, T, dtype=torch.long)
    for b in range(1, B):
        n = int(torch.randint(T // 3, T, (1,), generator=g))
        if padding == "right":
            am[b, n:] = 0
        else:
            am[b, :T - n] = 0
    pos = (am.cumsum(-1) - 1).clamp_min(0) if padding == "left" else torch.arange(T)[None].expand(B, T).clone()
    return x, am, pos


# ----------------------------------------------------------------------------------------------
# seeded synthetic weights (no pretrained weights / network exist: SURVEY.md §8d)
# ----------------------------------------------------------------------------------------------
def init_head_weights(hc: HeadConfig, seed: int = 1, dtype=torch.float32) -> Dict[str, Tensor]:
    """xavier-uniform Linear weights, zero biases, unit LayerNorms (tokenizer.py:59-72) under
    `torch.manual_seed(seed)`, with the reference's state-dict key names."""
    g = torch.Generator().manual_seed(seed)
    C, ff = hc.hidden_dim, hc.dim_feedforward

    def xavier(o, i):
        a = math.sqrt(6.0 / (i + o))
        return (torch.rand(o, i, generator=g, dtype=torch.float32) * 2 - 1).mul_(a).to(dtype)

    sd: Dict[str, Tensor] = {}
    for name, depth in (("inner_encoder", hc.inner_cluster_layers), ("inter_encoder", hc.intra_cluster_layers)):
        for nm in ("norm1", "norm2"):
            sd[f"{name}.{nm}.weight"] = torch.ones(C, dtype=dtype)
            sd[f"{name}.{nm}.bias"] = torch.zeros(C, dtype=dtype)
        for i in range(depth):
            sd[f"{name}.layers.{i}.1.qkv.weight"] = xavier(3 * C, C)
            sd[f"{name}.layers.{i}.1.qkv.bias"] = torch.zeros(3 * C, dtype=dtype)
            sd[f"{name}.layers.{i}.1.proj.weight"] = xavier(C, C)
            sd[f"{name}.layers.{i}.1.proj.bias"] = torch.zeros(C, dtype=dtype)
        sd[f"{name}.mlp.fc1.weight"] = xavier(ff, C)
        sd[f"{name}.mlp.fc1.bias"] = torch.zeros(ff, dtype=dtype)
        sd[f"{name}.mlp.fc2.weight"] = xavier(C, ff)
        sd[f"{name}.mlp.fc2.bias"] = torch.zeros(C, dtype=dtype)
    sd["out.weight"] = xavier(hc.token_feat_dim, C)
    sd["out.bias"] = torch.zeros(hc.token_feat_dim, dtype=dtype)
    return sd


def init_tower_weights(vc: VitConfig, seed: int = 0, dtype=torch.float32, perturb: float = 0.02) -> Dict[str, Tensor]:
    """Seeded random ViT weights with HF CLIPVisionModel key names (transformers 5.x spelling).
    Standard deviations follow HF CLIP's published init (factor 1: q/k/v and fc2 at
    C^-0.5 (2L)^-0.5, out_proj at C^-0.5, fc1 at (2C)^-0.5, patch/position embedding 0.02, class
    embedding C^-0.5) so that token features keep a realistic spread through the depth (with it the
    DPC-kNN scores of random images land around 0.1); biases and LayerNorm affine
    parameters are additionally perturbed by `perturb` so that parity tests exercise them."""
    g = torch.Generator().manual_seed(seed)
    C, I, L, p = vc.hidden_size, vc.intermediate_size, vc.num_hidden_layers, TOWER_PREFIX
    in_std = C ** -0.5 * (2 * L) ** -0.5
    out_std = C ** -0.5
    fc_std = (2 * C) ** -0.5

    def rn(*shape, s):
        return (torch.randn(*shape, generator=g, dtype=torch.float32) * s).to(dtype)

    sd = {p + "embeddings.class_embedding": rn(C, s=C ** -0.5),
          p + "embeddings.patch_embedding.weight": rn(C, vc.num_channels, vc.patch_size, vc.patch_size, s=0.02),
          p + "embeddings.position_embedding.weight": rn(vc.num_patches + 1, C, s=0.02),
          p + "pre_layrnorm.weight": 1 + rn(C, s=perturb), p + "pre_layrnorm.bias": rn(C, s=perturb),
          p + "post_layernorm.weight": torch.ones(C, dtype=dtype), p + "post_layernorm.bias": torch.zeros(C, dtype=dtype)}
    for i in range(L):
        q = p + f"encoder.layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj"):
            sd[q + f"self_attn.{nm}.weight"] = rn(C, C, s=in_std)
            sd[q + f"self_attn.{nm}.bias"] = rn(C, s=perturb)
        sd[q + "self_attn.out_proj.weight"] = rn(C, C, s=out_std)
        sd[q + "self_attn.out_proj.bias"] = rn(C, s=perturb)
        for nm in ("layer_norm1", "layer_norm2"):
            sd[q + nm + ".weight"] = 1 + rn(C, s=perturb)
            sd[q + nm + ".bias"] = rn(C, s=perturb)
        sd[q + "mlp.fc1.weight"] = rn(I, C, s=fc_std); sd[q + "mlp.fc1.bias"] = rn(I, s=perturb)
        sd[q + "mlp.fc2.weight"] = rn(C, I, s=in_std); sd[q + "mlp.fc2.bias"] = rn(C, s=perturb)
    return sd


def planted_features(N: int, C: int, m: int, seed: int = 0, centre_std: float = 2.0,
                     noise_std: float = 0.05) -> Tensor:
    """Piecewise-constant feature map with m Voronoi regions on the sqrt(N) grid:
    exercises the dynamic-k branch (score > threshold) at the default threshold."""
    g = torch.Generator().manual_seed(seed)
    h = int(math.sqrt(N))
    sites = torch.rand(m, 2, generator=g) * h
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(h, dtype=torch.float32), indexing="ij")
    pts = torch.stack([yy.reshape(-1), xx.reshape(-1)], dim=-1)
    region = torch.cdist(pts, sites).argmin(dim=-1)
    centres = torch.randn(m, C, generator=g) * centre_std
    return centres[region] + torch.randn(N, C, generator=g) * noise_std


def lm_loss(logits: Tensor, new_labels: Tensor, attention_mask: Optional[Tensor]) -> Tensor:
    """The language-model loss of SetokimLlamaForCausalLM.forward, src/model/language_model/setokim_llama.py:145-160, call for call:
    `logits.float()`, shift so that tokens < n predict n, keep the positions whose NEXT token is attended (`attention_mask[..., 1:] != 0`),
    `nn.CrossEntropyLoss()` (ignore_index -100, mean)."""
    logits = logits.float()                                                          # :147
    if attention_mask is not None:                                                   # :149
        shift_attention_mask = attention_mask[..., 1:]                               # :150
        shift_logits = logits[..., :-1, :][shift_attention_mask != 0].contiguous()   # :151
        shift_labels = new_labels[..., 1:][shift_attention_mask != 0].contiguous()   # :152
    else:
        shift_logits = logits[..., :-1, :].contiguous()                              # :154
        shift_labels = new_labels[..., 1:].contiguous()                              # :155
    loss_fct = torch.nn.CrossEntropyLoss()                                           # :157
    return loss_fct(shift_logits.view(-1, shift_logits.size(-1)), shift_labels.view(-1))      # :158-160


def lm_loss_inputs(seed, B, T, V, padding):
    """Seeded logits (bf16-representable), labels with IGNORE_INDEX stretches (the prompt part) and an attention mask with padding."""
    g = torch.Generator().manual_seed(seed)
    logits = (torch.randn(B, T, V, generator=g) * 3).bfloat16().float()
    labels = torch.randint(0, V, (B, T), generator=g)
    labels[:, : T // 3] = -100
    am = torch.ones(B, T, dtype=torch.long)
    if padding == "right":
        for b in range(B): am[b, T - 1 - b:] = 0
    elif padding == "left":
        for b in range(B): am[b, : b + 1] = 0
    return logits, labels, (None if padding == "none" else am)
