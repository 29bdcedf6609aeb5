"""Live pin of the oracle against the reference's own code (RAC).  Runs only where /root/reference
exists (the build container); skipped on the GPU box."""
import os

import pytest
import torch

import rac_harness as R
import setok_oracle as O

pytestmark = [pytest.mark.reference,
              pytest.mark.skipif(not R.reference_available(), reason="/root/reference not present")]

torch.set_num_threads(min(8, os.cpu_count() or 1))


@pytest.fixture(scope="module")
def tok():
    d = R.make_clip_dir(64, 3, 4, 128, 112, 14, seed=0)
    return R.build_reference_tokenizer(d, hidden_dim=64, token_feat_dim=96, dim_feedforward=128,
                                       min_cluster_num=8, threshold=0.5)


def _sd(tok):
    return O.normalise_tower_keys(dict(tok.state_dict()))


HC = O.HeadConfig(hidden_dim=64, token_feat_dim=96, min_cluster_num=8, threshold=0.5, nheads=2, dim_feedforward=128)
VC = O.VitConfig(64, 128, 3, 4, 112, 14)


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_head_bitwise_on_identical_features(tok, seed):
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn(64, 64, generator=g)
    noise = torch.rand(64, generator=g)
    s0 = R.rac_head_single(tok, feats, threshold=1e9, noise=noise, return_stages=True)["score"].reshape(-1)
    thr = float(s0.sort(descending=True).values[10:12].mean())
    for kw in (dict(), dict(threshold=thr), dict(k=5, threshold=thr)):
        ref = R.rac_head_single(tok, feats, noise=noise, return_stages=True, **kw)
        got = O.head_forward(_sd(tok), HC, feats, noise=noise, **kw)
        assert torch.equal(ref["x"], got.x)
        assert torch.equal(ref["index_down"], got.index_down)
        assert torch.equal(ref["idx_cluster"], got.idx_cluster)
        assert torch.equal(ref["score"], got.score)            # same formula, same torch kernels
        assert torch.equal(ref["tokens"], got.tokens)


def test_cdist_restatement_is_bitwise(tok):
    g = torch.Generator().manual_seed(9)
    for n, c in ((26, 8), (256, 1024), (576, 256)):
        x = torch.randn(n, c, generator=g)
        assert torch.equal(torch.cdist(x, x), O.pairwise_dist(x))


def test_tower_restatement_vs_hf(tok):
    g = torch.Generator().manual_seed(3)
    images = torch.randn(2, 3, 112, 112, generator=g)
    for sel in (-2, -1, 1):
        tok.image_feature_encoder.select_layer = sel
        ref = tok.image_feature_encoder(images)
        got = O.tower_forward(_sd(tok), VC, images, sel)
        torch.testing.assert_close(got, ref, rtol=1e-5, atol=5e-6)
    tok.image_feature_encoder.select_layer = -2


def test_pos_encoding_bitwise(tok):
    _, mod = R.load_reference()
    for h, w, c in ((16, 16, 1024), (24, 24, 1024), (14, 14, 768), (3, 5, 10)):
        pe = mod.PositionalEncoding2D(c)(torch.zeros(1, h, w, c))
        assert torch.equal(pe.reshape(h * w, c), O.pos_encoding_2d(h, w, c))


def test_reference_defects_still_present(tok):
    """D1/D2 of SURVEY.md §0.2 — documents why RAC needs its two repairs."""
    with pytest.raises(Exception):
        tok(torch.randn(2, 3, 112, 112))
    with pytest.raises(ValueError):
        tok.inter_encoder(torch.randn(5, 64))


def test_qformer_restatement_vs_reference_live():
    """a9: oracle.qformer_forward against the reference's BertEmbeddings + BertEncoder (module.py:151-690), live."""
    dc = O.DetokConfig(token_feat_dim=48, hidden_dim=32, patch_size=14, image_size=42, decoder_embed_dim=32, decoder_nheads=2,
                       decoder_depth=1, num_hidden_layers=3, cross_attention_freq=2, mapper_hidden=32, mapper_heads=4,
                       mapper_intermediate=64)
    sd = O.init_detok_weights(dc, seed=11)
    emb, enc = R.build_reference_qformer(hidden=32, heads=4, intermediate=64, layers=3, cross_freq=2, encoder_width=32,
                                         num_queries=dc.num_queries)
    torch.manual_seed(0)
    x = torch.randn(3, 6, 32)
    mask = torch.ones(3, 6); mask[0, 2:] = 0; mask[2, 5:] = 0
    q = sd["mask_tokens"].expand(3, -1, -1)
    ref = R.rac_qformer(emb, enc, sd, q, x, mask)
    got = O.qformer_forward(sd, dc, q, x, mask)
    torch.testing.assert_close(got, ref, rtol=1e-6, atol=1e-6)
    ref_nomask = R.rac_qformer(emb, enc, sd, q, x, None)
    torch.testing.assert_close(O.qformer_forward(sd, dc, q, x, None), ref_nomask, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("kw", [dict(), dict(padding_side="left"), dict(max_length=11), dict(max_length=5, padding_side="left")])
def test_splice_restatement_vs_reference_live(kw):
    """§8(f) row 1: oracle.splice_multimodal against the reference's prepare_inputs_labels_for_multimodal
    (setokim_arch.py:213-355) executed unmodified; every output tensor bit-equal, None conventions included."""
    ids, am, labels, feats, W = O.splice_inputs(21, 6, 14, 50, 12)
    pos = torch.arange(14).expand(6, 14).clone()
    for p_, a_, l_ in ((pos, am, labels), (None, None, None), (None, am.bool(), labels), (pos, am, None)):
        ref = R.rac_prepare_inputs(ids, p_, a_, l_, feats, W, **kw)
        got = O.splice_multimodal(ids, p_, a_, l_, feats, W, **kw)
        for r, g in zip(ref, got):
            assert (r is None) == (g is None)
            if r is not None:
                assert r.dtype == g.dtype and torch.equal(r, g)


def test_llama_restatement_vs_installed_hf_live():
    """cfg 5: oracle.llama_forward against the installed HuggingFace LlamaForCausalLM (third-party arithmetic the reference calls at
    setokim_llama.py:130-143), eager attention, fp32 — bit-equal on token positions."""
    from transformers import LlamaConfig, LlamaForCausalLM
    kw = dict(hidden_size=64, intermediate_size=176, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4, vocab_size=100)
    lc = O.LlamaConfigLite(**kw)
    sd = O.init_llama_weights(lc, seed=3)
    cfg = LlamaConfig(**kw, rms_norm_eps=lc.rms_norm_eps, rope_theta=lc.rope_theta, attention_bias=False, mlp_bias=False, tie_word_embeddings=False)
    cfg._attn_implementation = "eager"
    m = LlamaForCausalLM(cfg).eval()
    m.load_state_dict(sd, strict=True)
    for padding in ("right", "left"):
        x, am, pos = O.llama_inputs(lc, 5, 3, 13, padding)
        with torch.no_grad():
            ref = m(inputs_embeds=x, attention_mask=am, position_ids=pos).logits
        _, got = O.llama_forward(sd, lc, x, am, pos)
        assert torch.equal(got[am.bool()], ref[am.bool()])


def test_checkpoint_key_names_load_in_the_reference_and_back(tok, tmp_path):
    """A stage-1 checkpoint written by setok_amd.checkpoint loads in the reference's SetokTokenizer through the reference's own
    `get_w(weights, 'tokenizer')` + `load_state_dict(strict=False)` (setokim_arch.py:94-99), and a reference-side checkpoint loads here."""
    from setok_amd import SetokTokenizer, checkpoint as C
    torch.manual_seed(5)
    vc = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=4, image_size=112, patch_size=14)
    mine = SetokTokenizer(vision_tower=vc, hidden_dim=64, token_feat_dim=96, min_cluster_num=8, nheads=2, dim_feedforward=128)
    with torch.no_grad():
        for n, p in mine.named_parameters():
            if not n.startswith("image_feature_encoder."):
                p.copy_(torch.randn_like(p))
    path = tmp_path / "setok_stage1.bin"
    C.save_tokenizer_checkpoint(mine, path)

    import copy
    ref = copy.deepcopy(tok)
    weights = torch.load(path, map_location="cpu")
    get_w = lambda weights, keyword: {k.split(keyword + '.')[1]: v for k, v in weights.items() if keyword in k}      # as the reference spells it
    res = ref.load_state_dict(get_w(weights, "tokenizer"), strict=False)
    assert res.unexpected_keys == []
    assert all(k.startswith("image_feature_encoder.") for k in res.missing_keys)
    ref_sd, my_sd = ref.state_dict(), mine.state_dict()
    head = [k for k in my_sd if not k.startswith("image_feature_encoder.")]
    assert len(head) >= 34 and all(torch.equal(ref_sd[k], my_sd[k]) for k in head)

    # the other direction: the stage-1 model's state dict holds the tokenizer under `tokenizer.` (and a detokenizer beside it)
    theirs = {"tokenizer." + k: v.clone() for k, v in tok.state_dict().items() if not k.startswith("image_feature_encoder.")}
    theirs["detokenizer.decoder_norm.weight"] = torch.ones(3)             # comes along through the keyword match, dropped by strict=False
    res = C.load_pretrained_tokenizer(mine, theirs)
    assert res.unexpected_keys == ["decoder_norm.weight"]
    assert all(torch.equal(tok.state_dict()[k], mine.state_dict()[k]) for k in head)


def test_lm_loss_restatement_vs_reference_statements_live():
    """The oracle's lm_loss against the reference's own loss statements executed from the reference file (rac_harness.rac_lm_loss)."""
    for seed, B, T, V, padding in ((0, 2, 9, 37, "none"), (5, 4, 17, 101, "right"), (6, 3, 13, 64, "left")):
        logits, labels, am = O.lm_loss_inputs(seed, B, T, V, padding)
        assert torch.equal(R.rac_lm_loss(logits, labels, am), O.lm_loss(logits, labels, am))


def test_pixel_terms_equal_the_reference_losses():
    """§8(f) row 2: the oracle's pixel terms against the reference's own WeightedMSELoss module (loss/mse.py, loaded by file path) and the
    statements of the GAN loss's pixel term (loss/discriminator.py:161,170: torch.abs(inputs - reconstructions), torch.mean)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("rac_mse", os.path.join(R.REF_ROOT, "src", "model", "loss", "mse.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    g = torch.Generator().manual_seed(1)
    a, b = torch.randn(4, 3, 28, 42, generator=g), torch.randn(4, 3, 28, 42, generator=g)
    assert torch.equal(O.pixel_loss(a, b, "mse"), mod.WeightedMSELoss()(a, b))
    assert torch.equal(O.pixel_loss(a, b, "l1"), torch.mean(torch.abs(b.contiguous() - a.contiguous())))


def test_stage2_oracle_projector_and_splice_match_the_reference_autograd():
    """The stage-2 golden (tests/golden/stage2.npz) is generated by rac_stage2_grads; here the same quantities come from the ORACLE's projector +
    splice under torch autograd, live against the reference's modules: equal loss and gradients (fp32 rounding)."""
    Bm = R.load_reference_projector_builder()
    seed, B, T, V, Dt, Dh = 21, 4, 11, 30, 48, 32
    ids, am, labels, feats, W = O.splice_inputs(seed, B, T, V, Dh)
    g = torch.Generator().manual_seed(seed)
    toks = [torch.randn(f.shape[0], Dt, generator=g) for f in feats]
    w_down = torch.randn(V, Dh, generator=g) * 0.3
    torch.manual_seed(seed)
    proj = Bm.build_vision_projector("mlp2x_gelu", mm_hidden_size=Dt, hidden_size=Dh)
    pos = torch.arange(T).expand(B, T).clone()
    loss, embeds, new_labels, pg, tg, _ = R.rac_stage2_grads(proj, toks, ids, pos, am, labels, W, w_down)
    psd = {n: p.detach().clone().requires_grad_(True) for n, p in proj.named_parameters()}
    tl = [t.clone().requires_grad_(True) for t in toks]
    of = [O.projector_forward(psd, "mlp2x_gelu", t) for t in tl]
    _, _, oemb, olab = O.splice_multimodal(ids, pos, am, labels, of, W)
    ol = O.stage2_downstream(oemb, olab, w_down)
    ol.backward()
    assert torch.equal(olab, new_labels) and torch.allclose(oemb, embeds, rtol=1e-6, atol=1e-6) and torch.allclose(ol, loss, rtol=1e-6)
    for n in pg:
        assert torch.allclose(psd[n].grad, pg[n], rtol=1e-4, atol=1e-6), n
    for a, b in zip(tl, tg):
        assert torch.allclose(a.grad if a.grad is not None else torch.zeros_like(a), b, rtol=1e-4, atol=1e-6)
