"""The oracle (oracle/setok_oracle.py) against the committed golden vectors, which are outputs of
the REFERENCE itself (tests/golden/make_golden.py ran RAC in the build container).  CPU only."""
import os

import numpy as np
import pytest
import torch

import setok_oracle as O

torch.set_num_threads(min(8, os.cpu_count() or 1))


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


HEAD_CASES = ["fallback", "dynamic", "planted", "masked", "k_explicit", "n16_direct"]


@pytest.mark.parametrize("case", HEAD_CASES)
def test_head_small_matches_reference(golden_dir, case):
    z = _load(golden_dir, "head_small")
    cfg = dict(zip([str(k) for k in z["cfg_keys"]], z["cfg_vals"]))
    hc = O.HeadConfig(hidden_dim=int(cfg["hidden_dim"]), token_feat_dim=int(cfg["token_feat_dim"]),
                      min_cluster_num=int(cfg["min_cluster_num"]), threshold=float(cfg["threshold"]),
                      nheads=int(cfg["nheads"]), dim_feedforward=int(cfg["dim_feedforward"]))
    sd = {k[2:]: _t(z[k]) for k in z.files if k.startswith("w:")}
    k = int(z[f"{case}:k"]); thr = float(z[f"{case}:threshold"])
    tm = _t(z[f"{case}:token_mask"]) if f"{case}:token_mask" in z.files else None
    nz = _t(z[f"{case}:noise"]) if f"{case}:noise" in z.files else None
    r = O.head_forward(sd, hc, _t(z[f"{case}:feats"]), k=None if k < 0 else k,
                       threshold=None if thr < 0 else thr, token_mask=tm, noise=nz)
    assert torch.equal(r.x, _t(z[f"{case}:x"]))                       # pos-enc add is bit-exact
    assert torch.equal(r.index_down, _t(z[f"{case}:index_down"]))     # integers: bit-exact
    assert torch.equal(r.idx_cluster, _t(z[f"{case}:idx_cluster"]))
    assert r.idx_cluster.dtype == torch.int64 and tuple(r.score.shape) == (1, r.x.shape[0])
    torch.testing.assert_close(r.score, _t(z[f"{case}:score"]), rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(r.group, _t(z[f"{case}:group"]), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(r.tokens, _t(z[f"{case}:tokens"]), rtol=1e-4, atol=1e-5)


def test_cluster_full_dims_matches_reference(golden_dir):
    z = _load(golden_dir, "cluster_full")
    names = sorted({k.split(":")[0] for k in z.files})
    assert len(names) == 12
    for name in names:
        N, C, m, seed, k, mcn, thr = z[name + ":spec"]
        N, C, m, seed, k, mcn = int(N), int(C), int(m), int(seed), int(k), int(mcn)
        h = int(N ** 0.5)
        x = O.planted_features(N, C, m, seed=seed) + O.pos_encoding_2d(h, h, C)
        r = O.cluster_dpc_knn(x, k, float(thr), mcn)
        assert torch.equal(r.index_down, _t(z[name + ":index_down"]).long()), name
        assert torch.equal(r.idx_cluster, _t(z[name + ":idx_cluster"]).long()), name
        torch.testing.assert_close(r.score, _t(z[name + ":score"]), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("tag", ["sel-2_fallback", "sel-2_dynamic", "sel-1_fallback", "sel-1_dynamic"])
def test_e2e_small_matches_reference(golden_dir, tag):
    z = _load(golden_dir, "e2e_small")
    sd = {k[2:]: _t(z[k]) for k in z.files if k.startswith("w:")}
    vc = O.VitConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=4,
                     image_size=112, patch_size=14)
    sel = int(tag.split("_")[0][3:])
    hc = O.HeadConfig(hidden_dim=64, token_feat_dim=96, min_cluster_num=8, threshold=0.5, nheads=2,
                      dim_feedforward=128, mm_vision_select_layer=sel)
    thr = float(z[f"{tag}:threshold"])
    feats, res = O.encode(sd, vc, hc, _t(z["images"]), threshold=thr, noise=_t(z["noise"]))
    # the tower is third-party arithmetic (HF sdpa vs restated eager softmax): fp32 rounding class
    torch.testing.assert_close(feats, _t(z[f"{tag}:feats"]), rtol=1e-5, atol=5e-6)
    for i, r in enumerate(res):
        assert torch.equal(r.index_down, _t(z[f"{tag}:{i}:index_down"]))
        assert torch.equal(r.idx_cluster, _t(z[f"{tag}:{i}:idx_cluster"]))
        torch.testing.assert_close(r.tokens, _t(z[f"{tag}:{i}:tokens"]), rtol=1e-4, atol=2e-5)


def test_vitl_head_from_reference_features(golden_dir):
    """Full ViT-L dims: head on the reference's own tower features (identical fp32 input)."""
    z = _load(golden_dir, "vitl_224")
    hc = O.HeadConfig(threshold=0.125)
    sd = O.init_head_weights(hc, seed=int(z["spec"][1]))
    feats = _t(z["feats"])
    for i in range(feats.shape[0]):
        r = O.head_forward(sd, hc, feats[i])
        assert torch.equal(r.index_down, _t(z[f"{i}:index_down"]).long())
        assert torch.equal(r.idx_cluster, _t(z[f"{i}:idx_cluster"]).long())
        torch.testing.assert_close(r.score, _t(z[f"{i}:score"]), rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(r.group, _t(z[f"{i}:group"]), rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(r.tokens, _t(z[f"{i}:tokens"]), rtol=1e-4, atol=1e-5)
        fb = O.head_forward(sd, hc, feats[i], threshold=0.5)
        assert fb.index_down.numel() == 64
        assert torch.equal(fb.index_down, _t(z[f"{i}:fb:index_down"]).long())
        assert torch.equal(fb.idx_cluster, _t(z[f"{i}:fb:idx_cluster"]).long())
        torch.testing.assert_close(fb.tokens.double().sum(dim=1), _t(z[f"{i}:fb:tokens_rowsum"]), rtol=1e-4, atol=1e-3)


def test_vitl_tower_restatement_close_to_reference(golden_dir):
    """Third-party boundary: restated CLIP ViT-L/14 vs the reference's HF tower (seeded weights)."""
    z = _load(golden_dir, "vitl_224")
    vc = O.VitConfig()
    sd = O.init_tower_weights(vc, seed=int(z["spec"][0]))
    g = torch.Generator().manual_seed(int(z["spec"][2]))
    images = torch.randn(2, 3, 224, 224, generator=g)
    feats = O.tower_forward(sd, vc, images[:1], -2)
    torch.testing.assert_close(feats, _t(z["feats"])[:1], rtol=1e-4, atol=1e-4)


def test_cluster_properties_and_quirks():
    """Size-independent properties the reference's algorithm guarantees."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(100, 32, generator=g)
    r = O.cluster_dpc_knn(x, 5, 1e9, 7)
    assert r.fallback and r.index_down.numel() == 7
    assert torch.equal(r.index_down, torch.sort(r.index_down).values)              # sorted fallback centres
    assert torch.equal(r.idx_cluster[r.index_down], torch.arange(7))                # every centre owns itself
    assert int(r.idx_cluster.min()) >= 0 and int(r.idx_cluster.max()) < 7
    s = r.score.reshape(-1)
    thr = float((s.max() + s.sort().values[-2]) / 2)
    r2 = O.cluster_dpc_knn(x, 5, thr, 7)
    assert not r2.fallback and r2.index_down.numel() == 1 and int(r2.idx_cluster.max()) == 0
    # the highest-density token gets delta = min_j rowmax_j (the `[None, None]` broadcast quirk, :98-99)
    top = int(r.density.argmax())
    assert torch.isclose(r.delta[top], r.dist.max(dim=-1).values.min())


def test_pos_encoding_layout():
    pe = O.pos_encoding_2d(3, 4, 10)
    assert tuple(pe.shape) == (12, 10)
    ch = 6                                                                           # ceil(10/4)*2
    inv = 1.0 / (10000 ** (torch.arange(0, ch, 2).float() / ch))
    row2 = torch.stack(((2 * inv).sin(), (2 * inv).cos()), -1).flatten()
    col3 = torch.stack(((3 * inv).sin(), (3 * inv).cos()), -1).flatten()
    assert torch.allclose(pe[2 * 4 + 3, :ch], row2) and torch.allclose(pe[2 * 4 + 3, ch:], col3[:4])


def test_projector_types():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(5, 8, generator=g)
    psd = {"0.weight": torch.randn(6, 8, generator=g), "0.bias": torch.randn(6, generator=g),
           "2.weight": torch.randn(6, 6, generator=g), "2.bias": torch.randn(6, generator=g)}
    y = O.projector_forward(psd, "mlp2x_gelu", x)
    ref = torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(x, psd["0.weight"], psd["0.bias"])),
                                     psd["2.weight"], psd["2.bias"])
    assert torch.equal(y, ref)
    assert O.projector_forward({}, "identity", x) is x
    with pytest.raises(ValueError):
        O.projector_forward({}, "bogus", x)


def test_sensitivity_analysis_accepts_reference_outputs(golden_dir):
    """The perturbation analysis that decides which integer decisions parity tests must match
    bit-for-bit (oracle.cluster_sensitivity) must itself accept the reference's own fp32 outputs, and
    must flag the known rounding-level tie (top-32 fallback on planted features: the reference's fp32
    run and an exact fp64 run pick different 32nd centres)."""
    z = _load(golden_dir, "cluster_full")
    n_certain = 0
    for name in sorted({k.split(":")[0] for k in z.files}):
        N, C, m, seed, k, mcn, thr = z[name + ":spec"]
        N, C, m, seed, k, mcn = int(N), int(C), int(m), int(seed), int(k), int(mcn)
        if N != 256:
            continue
        x = O.planted_features(N, C, m, seed=seed) + O.pos_encoding_2d(16, 16, C)
        sens = O.cluster_sensitivity(x, k, float(thr), mcn)
        ref_c, ref_i = _t(z[name + ":index_down"]).long(), _t(z[name + ":idx_cluster"]).long()
        st = O.check_cluster_parity(ref_c, ref_i, ref_c, ref_i, sens)
        O.check_score(_t(z[name + ":score"]), sens)
        n_certain += st["centres_certain"]
        if "mcn32" in name:
            assert not sens["centres_certain"]
    assert n_certain == 5


# ---- a9: reconstruction decoder (cfg 3) ------------------------------------------------------------------------------
def _detok_case(golden_dir, name):
    z = _load(golden_dir, "detok")
    kw = {str(k): v for k, v in zip(z[name + ":cfg_keys"], z[name + ":cfg_vals"])}
    kw = {k: (float(v) if k == "mlp_ratio" else int(v)) for k, v in kw.items()}
    dc = O.DetokConfig(**kw)
    sd = O.init_detok_weights(dc, seed=int(z[name + ":seed"]))
    return dc, sd, _t(z[name + ":x"]), _t(z[name + ":mask"]), _t(z[name + ":mapped_ref"])


@pytest.mark.parametrize("name", ["small", "bertbase"])
def test_qformer_matches_reference(golden_dir, name):
    """The Q-Former restatement against the reference's BertEmbeddings + BertEncoder output (padded tokens + mask)."""
    dc, sd, x, mask, ref = _detok_case(golden_dir, name)
    st = O.detokenizer_forward(sd, dc, x, mask, return_stages=True)
    torch.testing.assert_close(st["mapped"], ref, rtol=1e-5, atol=1e-5)
    assert tuple(st["out"].shape) == (x.shape[0], dc.num_queries, dc.decoder_embed_dim)
    assert torch.isfinite(st["out"]).all()


def test_detokenizer_padded_equals_ragged(golden_dir):
    """(1 - m) * -10000 (module.py:849) == leaving the padded keys out: each image alone, unpadded, gives the same rows."""
    dc, sd, x, mask, _ = _detok_case(golden_dir, "small")
    full = O.detokenizer_forward(sd, dc, x, mask)
    for i in range(x.shape[0]):
        n = int(mask[i].sum())
        one = O.detokenizer_forward(sd, dc, x[i:i + 1, :n], None)
        torch.testing.assert_close(one[0], full[i], rtol=1e-5, atol=1e-5)
    # and padding VALUES are irrelevant
    x2 = x.clone(); x2[mask == 0] = 123.0
    torch.testing.assert_close(O.detokenizer_forward(sd, dc, x2, mask), full, rtol=1e-6, atol=1e-6)


def test_detokenizer_pos_table_width():
    dc = O.DetokConfig(token_feat_dim=8, hidden_dim=16, image_size=28, decoder_embed_dim=32, decoder_nheads=2, decoder_depth=1,
                       num_hidden_layers=1, mapper_hidden=16, mapper_heads=2, mapper_intermediate=32)
    sd = O.init_detok_weights(dc)
    with pytest.raises(ValueError):                      # the reference's `x + pos_emb` cannot broadcast 16 -> 32 channels
        O.detokenizer_forward(sd, dc, torch.randn(1, 3, 8), None)


# ---- §8(f) row 1: prepare_inputs_labels_for_multimodal ---------------------------------------------------------------
SPLICE_NAMES = ["right", "left", "trunc", "trunc_left", "nopad_long"]


def _splice_case(golden_dir, name):
    z = _load(golden_dir, "splice")
    seed, B, T, V, D, maxlen, left = [int(v) for v in z[name + ":spec"]]
    ids, am, labels, feats, W = O.splice_inputs(seed, B, T, V, D, pad=not name.startswith("nopad"))
    kw = dict(max_length=None if maxlen < 0 else maxlen, padding_side="left" if left else "right")
    return z, ids, am, labels, feats, W, kw


@pytest.mark.parametrize("name", SPLICE_NAMES)
def test_splice_matches_reference(golden_dir, name):
    """oracle.splice_multimodal against the outputs of the reference's own prepare_inputs_labels_for_multimodal: bit-exact."""
    z, ids, am, labels, feats, W, kw = _splice_case(golden_dir, name)
    T = ids.shape[1]
    pos = torch.arange(T).expand(ids.shape[0], T).clone()
    p, a, e, l = O.splice_multimodal(ids, pos, am, labels, feats, W, **kw)
    assert torch.equal(e, _t(z[f"{name}:full:embeds"])) and torch.equal(p, _t(z[f"{name}:full:pos"]))
    assert torch.equal(a, _t(z[f"{name}:full:mask"])) and torch.equal(l, _t(z[f"{name}:full:labels"]))
    assert a.dtype == am.dtype and not bool((l == O.TARGET_TOKEN_INDEX).any())
    p, a, e, l = O.splice_multimodal(ids, None, None, None, feats, W, **kw)
    assert p is None and a is None and l is None and torch.equal(e, _t(z[f"{name}:none:embeds"]))


def test_splice_needs_enough_images():
    ids, am, labels, feats, W = O.splice_inputs(7, 4, 10, 30, 8)
    with pytest.raises(IndexError):
        O.splice_multimodal(ids, None, am, labels, feats[:-1], W)


# ---- §8(f) row 4: gradients of the trainable head ----------------------------------------------------------------------
def test_head_param_grads_match_reference_autograd(golden_dir):
    """oracle.head_param_grads (autograd through the oracle's forward) against the gradients the REFERENCE's modules produce
    under the reference's own autograd (tests/golden/head_grads.npz)."""
    z = _load(golden_dir, "head_small")
    gz = _load(golden_dir, "head_grads")
    sd = {k[2:]: _t(z[k]) for k in z.files if k.startswith("w:")}
    hc = O.HeadConfig(hidden_dim=64, token_feat_dim=96, min_cluster_num=8, threshold=0.5, nheads=2, dim_feedforward=128)
    feats = [_t(z["dynamic:feats"]), _t(z["planted:feats"])]
    ups = [_t(gz["up:0"]), _t(gz["up:1"])]
    grads, res = O.head_param_grads(sd, hc, feats, ups, threshold=float(gz["threshold"]))
    assert [r.tokens.shape[0] for r in res] == gz["counts"].tolist()
    names = [k[2:] for k in gz.files if k.startswith("g:")]
    assert set(names) == set(grads) and len(names) == 34
    for n in names:
        torch.testing.assert_close(grads[n], _t(gz["g:" + n]), rtol=1e-5, atol=1e-6)


# ---- §8(f) last row: the LLM prefill of cfg 5 -------------------------------------------------------------------------------
LLAMA_NAMES = ["tiny_right", "tiny_left", "dh128", "dh128_left", "gqa_tiny_left", "gqa_dh128", "mqa_dh128_left"]     # the last three: num_key_value_heads < num_attention_heads


def _llama_case(golden_dir, name):
    z = _load(golden_dir, "llama")
    kw = {str(k): int(v) for k, v in zip(z[name + ":cfg_keys"], z[name + ":cfg_vals"])}
    lc = O.LlamaConfigLite(**kw)
    seed, B, T, left = [int(v) for v in z[name + ":spec"]]
    sd = O.init_llama_weights(lc, seed=seed)
    x, am, pos = O.llama_inputs(lc, seed, B, T, "left" if left else "right")
    return lc, sd, x, am, pos, _t(z[name + ":hidden"]), _t(z[name + ":logits"])


@pytest.mark.parametrize("name", LLAMA_NAMES)
def test_llama_prefill_matches_hf(golden_dir, name):
    """oracle.llama_forward against HuggingFace LlamaForCausalLM's outputs (the `self.model` + `self.lm_head` of setokim_llama.py:130-143)
    at every position that is a token (padded positions carry arbitrary values in HF and are masked out of the loss, :149-152)."""
    lc, sd, x, am, pos, hidden, logits = _llama_case(golden_dir, name)
    h, lg = O.llama_forward(sd, lc, x, am, pos)
    v = am.bool()
    torch.testing.assert_close(h[v], hidden[v], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(lg[v], logits[v], rtol=1e-5, atol=1e-5)


def test_lm_loss_matches_reference(golden_dir):
    """setokim_llama.py:145-160 — the oracle's restatement against the loss the reference's own statements gave (tests/golden/lm_loss.npz)."""
    z = np.load(os.path.join(golden_dir, "lm_loss.npz"))
    cases = sorted({k.split(":")[0] for k in z.files})
    assert len(cases) == 3
    for c in cases:
        seed, B, T, V = (int(v) for v in z[c + ":spec"])
        logits, labels, am = O.lm_loss_inputs(seed, B, T, V, str(z[c + ":padding"]))
        got = O.lm_loss(logits, labels, am)
        assert abs(float(got) - float(z[c + ":loss"][0])) <= 1e-6 * abs(float(z[c + ":loss"][0]))


def test_vitl336_head_from_reference_features(golden_dir):
    """BASELINE cfg4 dims (ViT-L/14-336: 576 patches): the oracle's head on the reference's own tower features (identical fp32 input)
    reproduces the reference's integers exactly and its tensors within fp32 rounding."""
    z = _load(golden_dir, "vitl_336")
    hc = O.HeadConfig(threshold=0.125)
    sd = O.init_head_weights(hc, seed=int(z["spec"][1]))
    feats = _t(z["feats"])
    assert tuple(feats.shape) == (2, 576, 1024)
    for i in range(feats.shape[0]):
        r = O.head_forward(sd, hc, feats[i])
        assert torch.equal(r.index_down, _t(z[f"{i}:index_down"]).long())
        assert torch.equal(r.idx_cluster, _t(z[f"{i}:idx_cluster"]).long())
        torch.testing.assert_close(r.score, _t(z[f"{i}:score"]), rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(r.group, _t(z[f"{i}:group"]), rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(r.tokens, _t(z[f"{i}:tokens"]), rtol=1e-4, atol=1e-5)


def test_vitl336_tower_restatement_close_to_reference(golden_dir):
    """Third-party boundary at T = 577: restated CLIP ViT-L/14-336 vs the reference's HF tower (seeded weights), one image."""
    z = _load(golden_dir, "vitl_336")
    vc = O.VitConfig(image_size=336)
    sd = O.init_tower_weights(vc, seed=int(z["spec"][0]))
    g = torch.Generator().manual_seed(int(z["spec"][2]))
    images = torch.randn(2, 3, 336, 336, generator=g)
    feats = O.tower_forward(sd, vc, images[:1], -2)
    torch.testing.assert_close(feats, _t(z["feats"])[:1], rtol=1e-4, atol=1e-4)


def test_llama_7b_dims_matches_hf(golden_dir):
    """BASELINE cfg5 at Vicuna-7B layer dims (hidden 4096, 32 x 128 heads, SwiGLU 11008, vocab 32000; two layers): the oracle's prefill
    against HuggingFace LlamaForCausalLM's stored outputs (weights regenerate from the seed)."""
    z = _load(golden_dir, "llama_7bdims")
    kw = {str(k): int(v) for k, v in zip(z["cfg_keys"], z["cfg_vals"])}
    seed, B, T, left = (int(v) for v in z["spec"])
    lc = O.LlamaConfigLite(**kw)
    sd = O.init_llama_weights(lc, seed=seed)
    x, am, pos = O.llama_inputs(lc, seed, B, T, "left" if left else "right")
    hidden, logits = O.llama_forward(sd, lc, x, am, pos)
    valid = am.bool()
    torch.testing.assert_close(hidden[valid], _t(z["hidden"])[valid], rtol=1e-5, atol=1e-5)
    stride = 32000 // _t(z["logits_cols"]).shape[-1] + 1
    torch.testing.assert_close(logits[:, :, ::stride][valid], _t(z["logits_cols"])[valid], rtol=1e-5, atol=1e-5)
    for b, t in enumerate(_t(z["last"]).tolist()):
        torch.testing.assert_close(logits[b, t], _t(z["logits_last"])[b], rtol=1e-5, atol=1e-5)


def test_pixel_head_restatement():
    """§8(f) row 2: the rearrangement of the pixel head against einops' own pattern, the pixel terms against torch's losses."""
    from einops import rearrange
    g = torch.Generator().manual_seed(0)
    B, gh, gw, p = 3, 4, 5, 7
    pt = torch.randn(B * gh * gw, p * p * 3, generator=g)
    want = rearrange(pt.reshape(B, gh * gw, p * p * 3), "n (h w) (p q c) -> n c (h p) (w q)", h=gh, w=gw, p=p, q=p, c=3)
    assert torch.equal(O.unpatchify(pt, B, gh, gw, p), want)
    a, b = torch.randn(B, 3, 28, 35, generator=g), torch.randn(B, 3, 28, 35, generator=g)
    assert torch.allclose(O.pixel_loss(a, b, "mse"), torch.nn.functional.mse_loss(a, b), rtol=1e-6)
    assert torch.allclose(O.pixel_loss(a, b, "l1"), torch.nn.functional.l1_loss(a, b), rtol=1e-6)


# ---- a second opinion on the ONE parity-unpinned sub-stage: the pixel decoder's timm Block (detokenizer.py:6,49-51) ---------------------------
def test_vit_block_restatement_against_an_independent_implementation():
    """`oracle.vit_block_forward` restates timm==0.9.16's `vision_transformer.Block` from its published algorithm (timm is not installable
    here, pyproject.toml:22): the restatement AND the HIP kernel were written by the same hand, so this sub-stage was single-sourced
    (VERDICT r03 weak 2).  HuggingFace `transformers` ships an independent implementation of the same block — `ViTLayer`: pre-LN, separate
    q / k / v projections, softmax(q k^T / sqrt(d_h)) v, output projection, residual; LayerNorm, Linear -> GELU (erf) -> Linear, residual.
    timm's fused `qkv` Linear is the row-wise concatenation [q; k; v] of HF's three.  This does NOT pin parity (it is not timm), but a
    misremembered block structure — post-LN, scale placement, tanh-GELU, a LayerScale, the head split order — would show here."""
    import torch
    from transformers import ViTConfig
    from transformers.models.vit.modeling_vit import ViTLayer
    import setok_oracle as O
    for hidden, heads, inter, eps, seed in ((64, 4, 256, 1e-6, 0), (96, 6, 192, 1e-5, 1)):
        torch.manual_seed(seed)
        cfg = ViTConfig(hidden_size=hidden, num_hidden_layers=1, num_attention_heads=heads, intermediate_size=inter, hidden_act="gelu",
                        layer_norm_eps=eps, qkv_bias=True, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, attn_implementation="eager")
        layer = ViTLayer(cfg).eval()
        with torch.no_grad():
            for prm in layer.parameters():                                        # biases and LayerNorm affine away from their 0 / 1 defaults
                prm.copy_(torch.randn_like(prm) * (0.3 if prm.dim() == 1 else hidden ** -0.5))
            layer.layernorm_before.weight.add_(1.0); layer.layernorm_after.weight.add_(1.0)
        h = layer.state_dict()
        p = "pixel_decoder.0."
        sd = {p + "norm1.weight": h["layernorm_before.weight"], p + "norm1.bias": h["layernorm_before.bias"],
              p + "attn.qkv.weight": torch.cat([h["attention.q_proj.weight"], h["attention.k_proj.weight"], h["attention.v_proj.weight"]], 0),
              p + "attn.qkv.bias": torch.cat([h["attention.q_proj.bias"], h["attention.k_proj.bias"], h["attention.v_proj.bias"]], 0),
              p + "attn.proj.weight": h["attention.o_proj.weight"], p + "attn.proj.bias": h["attention.o_proj.bias"],
              p + "norm2.weight": h["layernorm_after.weight"], p + "norm2.bias": h["layernorm_after.bias"],
              p + "mlp.fc1.weight": h["mlp.fc1.weight"], p + "mlp.fc1.bias": h["mlp.fc1.bias"],
              p + "mlp.fc2.weight": h["mlp.fc2.weight"], p + "mlp.fc2.bias": h["mlp.fc2.bias"]}
        x = torch.randn(3, 17, hidden)
        with torch.no_grad():
            want = layer(x)
            want = want[0] if isinstance(want, (tuple, list)) else want
        got = O.vit_block_forward(sd, p, x, heads, eps)
        torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
        # and the comparison can fail: the same weights with the head split transposed (heads-major vs qkv-major) or tanh-GELU do not match
        bad = dict(sd); bad[p + "attn.qkv.weight"] = sd[p + "attn.qkv.weight"].reshape(3, heads, hidden // heads, hidden).transpose(0, 1).reshape(3 * hidden, hidden)
        assert (O.vit_block_forward(bad, p, x, heads, eps) - want).abs().max() > 1e-2
