"""RAC harness — "reference-as-composed" (SURVEY.md §0.2 / §8c).  TEST INFRASTRUCTURE ONLY.

Imports the *reference's own* modules from /root/reference (read-only, never copied) so that the
oracle restatement (oracle/setok_oracle.py) can be pinned against the reference's real arithmetic
and so that golden vectors (tests/golden/) can be generated.  This file only runs inside the build
container: /root/reference does not exist on the GPU box, and nothing under tests/ -m gpu, smoke()
or bench.py imports it.

Two import shims are needed because the reference pins older third-party packages
(pyproject.toml:15-26) than the ones installed here:
  * shim 1: `timm.models.layers.DropPath` is imported (tokenizer.py:7, module.py:7) but never
    instantiated when drop_path == 0.0 (module.py:82) -> identity stub module.
  * shim 2: module.py:16-21 imports `apply_chunking_to_forward`, `prune_linear_layer`,
    `find_pruneable_heads_and_indices` from transformers.modeling_utils (moved / removed upstream).

The harness supplies exactly two repairs to make the reference's forward composable, both
documented in SURVEY.md §0.2:
  * D1: SetokTokenizer.forward is only shape-consistent for ONE image's (N, C) features, so the
    batch is a python loop over images;
  * D2: inter_encoder needs a (1, L, C) input -> one unsqueeze / squeeze.
Everything else (CLIPVisionTower.forward, PositionalEncoding2D.forward, cluster_dpc_knn,
group_encoding, Block.forward, out) is the reference's code executed unmodified.
"""
from __future__ import annotations

import importlib.machinery
import importlib.util
import math
import os
import sys
import tempfile
import types

import torch

REF_ROOT = os.environ.get("SETOK_REFERENCE_ROOT", "/root/reference")
_SETOK_DIR = os.path.join(REF_ROOT, "src", "model", "setok")
_PKG = "_rac_setok"


def reference_available() -> bool:
    return os.path.isfile(os.path.join(_SETOK_DIR, "tokenizer.py"))


def _install_shims() -> None:
    if "timm" not in sys.modules:
        class DropPath(torch.nn.Identity):
            def __init__(self, drop_prob: float = 0.0, *a, **k):
                super().__init__()
        for name in ("timm", "timm.models", "timm.models.layers"):
            m = types.ModuleType(name)
            m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None, is_package=True)
            m.__path__ = []
            sys.modules[name] = m
        sys.modules["timm"].models = sys.modules["timm.models"]
        sys.modules["timm.models"].layers = sys.modules["timm.models.layers"]
        sys.modules["timm.models.layers"].DropPath = DropPath
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu
    for name in ("apply_chunking_to_forward", "prune_linear_layer"):
        if not hasattr(mu, name):
            setattr(mu, name, getattr(pu, name))
    if not hasattr(mu, "find_pruneable_heads_and_indices"):
        def find_pruneable_heads_and_indices(*a, **k):
            raise NotImplementedError("removed upstream; only used by prune_heads (module.py:397-405)")
        mu.find_pruneable_heads_and_indices = find_pruneable_heads_and_indices


def load_reference():
    """Load utils/module/clip_encoder/tokenizer from the reference by file path as a synthetic
    package (bypasses setok/__init__.py, which drags in timm ViT / diffusers / diffdist)."""
    if _PKG + ".tokenizer" in sys.modules:
        return sys.modules[_PKG + ".tokenizer"], sys.modules[_PKG + ".module"]
    if not reference_available():
        raise FileNotFoundError(f"reference not found under {REF_ROOT}")
    _install_shims()
    pkg = types.ModuleType(_PKG)
    pkg.__path__ = [_SETOK_DIR]
    pkg.__spec__ = importlib.machinery.ModuleSpec(_PKG, loader=None, is_package=True)
    sys.modules[_PKG] = pkg
    mods = {}
    for name in ("utils", "module", "clip_encoder", "tokenizer"):
        spec = importlib.util.spec_from_file_location(f"{_PKG}.{name}", os.path.join(_SETOK_DIR, f"{name}.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"{_PKG}.{name}"] = mod
        spec.loader.exec_module(mod)
        mods[name] = mod
    return mods["tokenizer"], mods["module"]


def make_clip_dir(hidden: int, layers: int, heads: int, mlp: int, image_size: int, patch: int,
                  seed: int = 0, out_dir: str | None = None) -> str:
    """Seeded random-init HF CLIPVisionModel saved to a directory AutoModel/AutoProcessor can load
    (no pretrained weights and no network exist here)."""
    from transformers import CLIPImageProcessor, CLIPVisionConfig, CLIPVisionModel
    out_dir = out_dir or tempfile.mkdtemp(prefix="rac_clip_siglip_")  # builder.py:19 wants 'siglip' in the name
    cfg = CLIPVisionConfig(hidden_size=hidden, intermediate_size=mlp, num_hidden_layers=layers,
                           num_attention_heads=heads, image_size=image_size, patch_size=patch,
                           projection_dim=hidden)
    torch.manual_seed(seed)
    model = CLIPVisionModel(cfg).eval()
    model.save_pretrained(out_dir)
    CLIPImageProcessor(size={"shortest_edge": image_size},
                       crop_size={"height": image_size, "width": image_size}).save_pretrained(out_dir)
    return out_dir


def build_reference_tokenizer(clip_dir: str, *, hidden_dim: int, token_feat_dim: int, nheads: int = 2,
                              dim_feedforward: int = 4096, min_cluster_num: int = 64,
                              threshold: float = 0.5, select_layer: int = -2, head_seed: int = 1,
                              inner_cluster_layers: int = 2, intra_cluster_layers: int = 2):
    tok_mod, _ = load_reference()
    torch.manual_seed(head_seed)
    tok = tok_mod.SetokTokenizer(vision_tower=clip_dir, mm_vision_select_layer=select_layer,
                                 hidden_dim=hidden_dim, token_feat_dim=token_feat_dim, nheads=nheads,
                                 dim_feedforward=dim_feedforward, min_cluster_num=min_cluster_num,
                                 threshold=threshold, inner_cluster_layers=inner_cluster_layers,
                                 intra_cluster_layers=intra_cluster_layers)
    return tok.eval()


class FixedNoise:
    """Context manager replacing torch.rand (tokenizer.py:91) by a caller-supplied noise vector so
    the density tie-break noise is an explicit, recorded input instead of global RNG state."""

    def __init__(self, noise: torch.Tensor | None):
        self.noise = noise

    def __enter__(self):
        self._orig = torch.rand
        noise = self.noise

        def fake_rand(*shape, **kw):
            if len(shape) == 1 and not isinstance(shape[0], int):
                shape = tuple(shape[0])
            if noise is None:
                return torch.zeros(*shape, dtype=kw.get("dtype", torch.float32))
            assert tuple(shape) == tuple(noise.shape), (shape, noise.shape)
            return noise.to(kw.get("dtype", torch.float32)).clone()
        torch.rand = fake_rand
        return self

    def __exit__(self, *exc):
        torch.rand = self._orig
        return False


@torch.no_grad()
def rac_head_single(tok, feats_nc: torch.Tensor, k=None, threshold=None, token_mask=None,
                    noise: torch.Tensor | None = None, return_stages: bool = False):
    """tokenizer.py:162-180 on ONE image's (N, C) tower features (repair D1), with the D2 unsqueeze."""
    x = feats_nc.unsqueeze(0)                                   # tokenizer.py:162
    B, hw, C = x.shape                                          # :163
    h = w = int(math.sqrt(x.shape[1]))                          # :164
    pos_emb = tok.position_embedding(x.reshape(B, h, w, C))     # :165-166 (rearrange == reshape)
    x = x + pos_emb.reshape(B, hw, C)                           # :167-168
    x = x.squeeze(0)                                            # :169
    _threshold = threshold if threshold else tok.threshold     # :171
    _k = k if k else tok.min_cluster_num                        # :172
    with FixedNoise(noise):
        index_down, idx_cluster, score = tok.cluster_dpc_knn(x, _k, token_mask, _threshold)  # :174
    centers = x[index_down, :]                                  # :177
    group = tok.group_encoding(x, centers, idx_cluster)         # :178
    inter = tok.inter_encoder(group.unsqueeze(0)).squeeze(0)    # :179 + repair D2
    out = tok.out(inter)                                        # :180
    if return_stages:
        return dict(x=x, index_down=index_down, idx_cluster=idx_cluster, score=score,
                    group=group, inter=inter, tokens=out)
    return out, idx_cluster, score


@torch.no_grad()
def rac_forward(tok, images: torch.Tensor, k=None, threshold=None, noise=None, return_stages=False):
    """Tower on the batch (clip_encoder.py:50-62), then the per-image head loop."""
    feats = tok.image_feature_encoder(images)                   # tokenizer.py:161
    outs = []
    for i in range(feats.shape[0]):
        nz = None if noise is None else noise[i]
        outs.append(rac_head_single(tok, feats[i], k=k, threshold=threshold, noise=nz,
                                    return_stages=return_stages))
    return feats, outs


# ----------------------------------------------------------------------------------------------
# a9 — the Q-Former of the reconstruction decoder.  `BertModel` itself does not construct on the installed
# transformers (init_weights -> all_tied_weights_keys) and `SetokDeTokenizer` needs timm, diffusers
# and a network fetch of bert-base-uncased (detokenizer.py:6,10,80), so the reference arithmetic that CAN run here is
# `BertEmbeddings` + `BertEncoder` (module.py:151-206, 586-690), driven exactly as BertModel.forward drives them
# (module.py:913-998: query_embeds only, all-ones self mask, inverted encoder mask).
# ----------------------------------------------------------------------------------------------
def build_reference_qformer(*, hidden: int, heads: int, intermediate: int, layers: int, cross_freq: int,
                            encoder_width: int, num_queries: int, eps: float = 1e-12):
    from transformers.models.bert import BertConfig
    _, module = load_reference()
    cfg = BertConfig(hidden_size=hidden, num_attention_heads=heads, intermediate_size=intermediate,
                     num_hidden_layers=layers, layer_norm_eps=eps)
    cfg.encoder_width = encoder_width            # detokenizer.py:82
    cfg.add_cross_attention = True               # :84
    cfg.cross_attention_freq = cross_freq        # :86
    cfg.query_length = num_queries               # :87
    emb = module.BertEmbeddings(cfg).eval()
    enc = module.BertEncoder(cfg).eval()
    for layer in enc.layer:                      # detokenizer.py:94-96
        layer.output = None
        layer.intermediate = None
    return emb, enc


@torch.no_grad()
def rac_qformer(emb, enc, sd, query_embeds: torch.Tensor, encoder_hidden_states: torch.Tensor,
                encoder_attention_mask: torch.Tensor | None, prefix: str = "mapper."):
    """Load `sd`'s `mapper.*` entries into the reference modules and run BertModel.forward's data path."""
    esd = {k[len(prefix + "embeddings."):]: v for k, v in sd.items() if k.startswith(prefix + "embeddings.")}
    missing = emb.load_state_dict(esd, strict=False)
    assert not missing.unexpected_keys, missing
    lsd = {k[len(prefix + "encoder."):]: v for k, v in sd.items() if k.startswith(prefix + "encoder.")}
    res = enc.load_state_dict(lsd, strict=True)
    B, Q, _ = query_embeds.shape
    x = emb(query_embeds=query_embeds)                                           # module.py:913-918
    ext = (1.0 - torch.ones(B, Q))[:, None, None, :] * -10000.0                  # :923-939, 848-849
    if encoder_attention_mask is None:
        encoder_attention_mask = torch.ones(encoder_hidden_states.shape[:2])     # :967-968
    enc_ext = (1.0 - encoder_attention_mask.to(x.dtype))[:, None, None, :] * -10000.0   # invert_attention_mask
    out = enc(x, attention_mask=ext, head_mask=[None] * len(enc.layer), encoder_hidden_states=encoder_hidden_states,
              encoder_attention_mask=enc_ext, return_dict=True, query_length=Q)   # :984-996
    return out.last_hidden_state


# ----------------------------------------------------------------------------------------------
# §8(f) row 1 — the reference's own prepare_inputs_labels_for_multimodal (setokim_arch.py:213-355), unmodified, on a host
# object that supplies what the method reads: get_vision_tower() (non-None), encode_images() (returns the per-image token
# matrices handed in — the encoder itself is rows a1-a8), get_model().embed_tokens, config, device.
# ----------------------------------------------------------------------------------------------
def load_reference_arch():
    name = "rac_src_model.setokim_arch"
    if name in sys.modules:
        return sys.modules[name]
    model_dir = os.path.join(REF_ROOT, "src", "model")
    for pkg, path in (("src", os.path.join(REF_ROOT, "src")), ("rac_src_model", model_dir)):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = [path]
            m.__spec__ = importlib.machinery.ModuleSpec(pkg, loader=None, is_package=True)
            sys.modules[pkg] = m
    spec = importlib.util.spec_from_file_location("src.constants", os.path.join(REF_ROOT, "src", "constants.py"))
    const = importlib.util.module_from_spec(spec); sys.modules["src.constants"] = const; spec.loader.exec_module(const)
    # the builders / loss imported at module top (setokim_arch.py:21-24) are not used by the method under test
    for sub, names in (("multimodal_encoder.builder", ["build_vision_tower"]), ("multimodal_projector.builder", ["build_vision_projector"]),
                       ("multimodal_generator.builder", ["build_vision_generator"]), ("loss", ["DiffLoss"])):
        parts = sub.split(".")
        for i in range(1, len(parts) + 1):
            full = "rac_src_model." + ".".join(parts[:i])
            if full not in sys.modules:
                m = types.ModuleType(full); m.__path__ = []
                m.__spec__ = importlib.machinery.ModuleSpec(full, loader=None, is_package=True)
                sys.modules[full] = m
        for n in names:
            setattr(sys.modules["rac_src_model." + sub], n, None)
    spec = importlib.util.spec_from_file_location(name, os.path.join(model_dir, "setokim_arch.py"))
    mod = importlib.util.module_from_spec(spec); sys.modules[name] = mod; spec.loader.exec_module(mod)
    return mod


@torch.no_grad()
def rac_prepare_inputs(input_ids, position_ids, attention_mask, labels, image_features, embed_weight,
                       max_length=None, padding_side="right"):
    arch = load_reference_arch()
    emb = torch.nn.Embedding.from_pretrained(embed_weight, freeze=True)

    class _Model:
        embed_tokens = emb

    class _Cfg:
        pass
    cfg = _Cfg()
    if max_length is not None:
        cfg.tokenizer_model_max_length = max_length
    cfg.tokenizer_padding_side = padding_side

    class Host(arch.SetokimMetaForCausalLM):
        config = cfg
        device = embed_weight.device

        def get_model(self):
            return _Model()

        def get_vision_tower(self):
            return object()

        def encode_images(self, images):
            return image_features

    images = torch.zeros(len(image_features), 3, 2, 2)          # only `images is None` / `.ndim` are looked at (:219-227)
    out = Host().prepare_inputs_labels_for_multimodal(input_ids, position_ids, attention_mask, None, labels, images)
    _, pos, am, _, embeds, new_labels = out
    return pos, am, embeds, new_labels


# ----------------------------------------------------------------------------------------------
# §8(f) row 4 — parameter gradients of the reference's head modules by the reference's own autograd:
# tokenizer.py:162-180 per image (as rac_head_single, but with gradients enabled; cluster_dpc_knn keeps its own
# torch.no_grad, tokenizer.py:79), L = sum_i <tokens_i, upstream_i>.
# ----------------------------------------------------------------------------------------------
def rac_head_grads(tok, feats_list, upstream_list, k=None, threshold=None, noise_list=None):
    tok.zero_grad(set_to_none=True)
    for p in tok.parameters():
        p.requires_grad_(True)
    loss = 0.0
    counts = []
    for i, feats_nc in enumerate(feats_list):
        x = feats_nc.detach().unsqueeze(0)
        B, hw, C = x.shape
        h = w = int(math.sqrt(hw))
        x = x + tok.position_embedding(x.reshape(B, h, w, C)).reshape(B, hw, C)
        x = x.squeeze(0)
        _threshold = threshold if threshold else tok.threshold
        _k = k if k else tok.min_cluster_num
        with FixedNoise(None if noise_list is None else noise_list[i]):
            index_down, idx_cluster, score = tok.cluster_dpc_knn(x, _k, None, _threshold)
        group = tok.group_encoding(x, x[index_down, :], idx_cluster)
        out = tok.out(tok.inter_encoder(group.unsqueeze(0)).squeeze(0))
        counts.append(out.shape[0])
        loss = loss + (out * upstream_list[i]).sum()
    loss.backward()
    grads = {n: p.grad.detach().clone() for n, p in tok.named_parameters()
             if p.grad is not None and not n.startswith("image_feature_encoder")}
    return grads, counts


# ----------------------------------------------------------------------------------------------
# Stage 2 (scripts/pretrain_mm_proj.sh:40, train_setokim.py:336-339): mm_in_projector is trained THROUGH the splice.  The reference's own
# `build_vision_projector` module (multimodal_projector/builder.py:33-59, loaded by file path) feeds the reference's own
# prepare_inputs_labels_for_multimodal (as above, but with gradients enabled), and torch autograd carries a downstream loss back.
# ----------------------------------------------------------------------------------------------
def load_reference_projector_builder():
    name = "rac_projector_builder"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF_ROOT, "src", "model", "multimodal_projector", "builder.py"))
    mod = importlib.util.module_from_spec(spec); sys.modules[name] = mod; spec.loader.exec_module(mod)
    return mod


def rac_stage2_grads(projector, tokens_list, input_ids, position_ids, attention_mask, labels, embed_weight, w_down,
                     max_length=None, padding_side="right", train_embed=False):
    """loss = downstream(prepare_inputs_labels_for_multimodal(..., encode_images = projector(tokens_i))).  Returns (loss, embeds, new_labels,
    {projector parameter gradients}, [d loss / d tokens_i], d loss / d embed_weight or None) — all by the reference's modules under torch autograd."""
    arch = load_reference_arch()
    emb = torch.nn.Embedding.from_pretrained(embed_weight.clone(), freeze=not train_embed)

    class _Model:
        embed_tokens = emb

    class _Cfg:
        pass
    cfg = _Cfg()
    if max_length is not None:
        cfg.tokenizer_model_max_length = max_length
    cfg.tokenizer_padding_side = padding_side
    toks = [t.detach().clone().requires_grad_(True) for t in tokens_list]
    for p in projector.parameters():
        p.requires_grad_(True)
    projector.zero_grad(set_to_none=True)

    class Host(arch.SetokimMetaForCausalLM):
        config = cfg
        device = embed_weight.device

        def get_model(self):
            return _Model()

        def get_vision_tower(self):
            return object()

        def encode_images(self, images):                        # setokim_arch.py:206-211 with the tokenizer's output handed in
            return [projector(t) for t in toks]

    images = torch.zeros(len(toks), 3, 2, 2)
    _, pos, am, _, embeds, new_labels = Host().prepare_inputs_labels_for_multimodal(input_ids, position_ids, attention_mask, None, labels, images)
    from setok_oracle import stage2_downstream
    loss = stage2_downstream(embeds, new_labels, w_down)
    loss.backward()
    pg = {n: p.grad.detach().clone() for n, p in projector.named_parameters()}
    tg = [t.grad.detach().clone() if t.grad is not None else torch.zeros_like(t) for t in toks]
    eg = emb.weight.grad.detach().clone() if train_embed else None
    return loss.detach(), embeds.detach(), new_labels, pg, tg, eg


def rac_lm_loss(logits, new_labels, attention_mask):
    """The reference's own loss statements (src/model/language_model/setokim_llama.py:145-160, inside `SetokimLlamaForCausalLM.forward`,
    which cannot be instantiated here) executed as they stand: the lines are read from the reference file at call time, dedented and run
    with `logits`, `new_labels`, `attention_mask`, `labels` bound.  Nothing of the file is kept in this repository."""
    import textwrap
    import torch
    from torch import nn
    path = os.path.join(REF_ROOT, "src", "model", "language_model", "setokim_llama.py")
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.strip() == "loss = None")
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("diff_loss_list"))
    code = textwrap.dedent("\n".join(lines[start:end]))
    ns = {"logits": logits, "new_labels": new_labels, "attention_mask": attention_mask, "labels": new_labels, "nn": nn, "torch": torch}
    exec(compile(code, path, "exec"), ns)
    return ns["loss"]
