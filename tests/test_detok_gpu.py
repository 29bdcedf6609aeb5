"""Parity of the reconstruction decoder (BASELINE config 3) on a real MI355X: setok_amd.SetokDeTokenizer,
through the C ABI, against (i) the REFERENCE's Q-Former output held in tests/golden/detok.npz and (ii) the CPU oracle for the
stages behind it (the timm pixel decoder is third-party code that is not installed — restated, see oracle).  `pytest -m gpu`."""
import os

import numpy as np
import pytest
import torch

import setok_oracle as O

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from setok_amd import SetokDeTokenizer
    from setok_amd.tokenizer import RaggedTokens

DEV = "cuda"
TOL = 1e-4


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _rel(got, ref):
    got, ref = got.detach().double().cpu(), ref.double().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def _case(golden_dir, name):
    z = np.load(os.path.join(golden_dir, "detok.npz"))
    kw = {str(k): v for k, v in zip(z[name + ":cfg_keys"], z[name + ":cfg_vals"])}
    kw = {k: (float(v) if k == "mlp_ratio" else int(v)) for k, v in kw.items()}
    dc = O.DetokConfig(**kw)
    sd = O.init_detok_weights(dc, seed=int(z[name + ":seed"]))
    return dc, sd, _t(z[name + ":x"]), _t(z[name + ":mask"]), _t(z[name + ":mapped_ref"])


def _build(dc, sd, dtype=torch.float32):
    det = SetokDeTokenizer(token_feat_dim=dc.token_feat_dim, hidden_dim=dc.hidden_dim, patch_size=dc.patch_size,
                           image_size=dc.image_size, decoder_embed_dim=dc.decoder_embed_dim, decoder_nheads=dc.decoder_nheads,
                           decoder_depth=dc.decoder_depth, mlp_ratio=dc.mlp_ratio,
                           feature_mapper_path_or_name=dict(hidden_size=dc.mapper_hidden, num_attention_heads=dc.mapper_heads,
                                                            intermediate_size=dc.mapper_intermediate, layer_norm_eps=dc.mapper_eps),
                           num_hidden_layers=dc.num_hidden_layers, cross_attention_freq=dc.cross_attention_freq)
    res = det.load_state_dict(sd, strict=False)          # the reference's key names: nothing extra, only the (recomputed) buffer missing
    assert not res.unexpected_keys and set(res.missing_keys) <= {"position_embedding.inv_freq"}
    return det.to(device=DEV, dtype=dtype).eval()


@pytest.mark.parametrize("name", ["small", "bertbase"])
def test_detokenizer_fp32_parity(golden_dir, name):
    dc, sd, x, mask, mapped_ref = _case(golden_dir, name)
    det = _build(dc, sd)
    st = det(x.to(DEV), mask.to(DEV), return_stages=True)
    assert _rel(st["mapped"], mapped_ref) < TOL           # against the REFERENCE's Q-Former (BertEmbeddings + BertEncoder)
    ora = O.detokenizer_forward(sd, dc, x, mask, return_stages=True)
    assert _rel(st["dec_in"], ora["dec_in"]) < TOL
    assert _rel(st["out"], ora["out"]) < TOL
    assert tuple(st["out"].shape) == (x.shape[0], dc.num_queries, dc.decoder_embed_dim)


def test_detokenizer_input_forms(golden_dir):
    """Padded + mask, RaggedTokens, a list of per-image tensors: identical results (bit-exact — the same kernels run)."""
    dc, sd, x, mask, _ = _case(golden_dir, "small")
    det = _build(dc, sd)
    a = det(x.to(DEV), mask.to(DEV))
    counts = mask.sum(1).long().tolist()
    rows = [x[i, :c].to(DEV) for i, c in enumerate(counts)]
    b = det(RaggedTokens(torch.cat(rows, 0), counts))
    c = det(rows)
    assert torch.equal(a, b) and torch.equal(a, c)
    # one image alone == its rows in the batch (no cross-image term anywhere)
    one = det([rows[2]])
    assert torch.equal(one[0], a[2])
    # no mask = all tokens
    full = det(x.to(DEV))
    ora = O.detokenizer_forward(sd, dc, x, None)
    assert _rel(full, ora) < TOL


def test_detokenizer_bf16_agreement(golden_dir):
    dc, sd, x, mask, mapped_ref = _case(golden_dir, "bertbase")
    det = _build(dc, sd, torch.bfloat16)
    st = det(x.to(DEV), mask.to(DEV), return_stages=True)
    assert st["out"].dtype == torch.bfloat16
    assert _rel(st["mapped"].float(), mapped_ref) < 3e-2  # bf16 throughput mode: documented tolerance, not the parity bar
    ora = O.detokenizer_forward(sd, dc, x, mask)
    assert _rel(st["out"].float(), ora) < 5e-2


def test_detokenizer_errors():
    with pytest.raises(ValueError):                       # the reference's own default (hidden_dim=4096 into LayerNorm(768)) cannot run
        SetokDeTokenizer()
    with pytest.raises(ValueError):
        SetokDeTokenizer(hidden_dim=768, decoder_embed_dim=4096)
    with pytest.raises(ValueError):
        SetokDeTokenizer(hidden_dim=768, decoder_embed_dim=768, feature_mapper_path_or_name="some/hub-model")
    det = SetokDeTokenizer(token_feat_dim=32, hidden_dim=64, image_size=28, decoder_embed_dim=64, decoder_nheads=2, decoder_depth=1,
                           num_hidden_layers=1, feature_mapper_path_or_name=dict(hidden_size=64, num_attention_heads=2,
                                                                                 intermediate_size=64)).to(DEV)
    with pytest.raises(ValueError):
        det(torch.zeros(2, 3, 16, device=DEV))
    with pytest.raises(ValueError):
        det(torch.zeros(2, 3, 32, device=DEV), torch.tensor([[1, 1, 0], [0, 0, 0]], device=DEV))


def test_detokenizer_full_dims_batch_invariance_bf16():
    """cfg3 dims (token_feat_dim 4096 -> Q-Former 768 / 12 heads / 6 layers, 324 queries, 16 ViT blocks of 768 / 16 heads), bf16, 24 images
    with ~36 tokens each: an image decoded alone is bit-identical to the same image inside the batch, and two runs are bit-identical."""
    det = SetokDeTokenizer(token_feat_dim=4096, hidden_dim=768, patch_size=14, image_size=256, decoder_embed_dim=768, decoder_nheads=16,
                           decoder_depth=16).to(device=DEV, dtype=torch.bfloat16).eval()
    g = torch.Generator().manual_seed(4)
    counts = [int(c) for c in torch.randint(24, 56, (24,), generator=g)]
    toks = [torch.randn(c, 4096, generator=g).to(device=DEV, dtype=torch.bfloat16) for c in counts]
    full = det(toks)
    assert tuple(full.shape) == (24, 324, 768) and bool(torch.isfinite(full.float()).all())
    assert torch.equal(full, det(toks))
    for i in (0, 7, 23):
        assert torch.equal(det([toks[i]])[0], full[i]), i
    sub = det(toks[5:9])
    assert torch.equal(sub, full[5:9])


# ---- the pixel head: the output the reference never defines -----------------------------------------------------------
@pytest.mark.parametrize("dt,tol,case", [(torch.float32, 1e-4, "small"), (torch.float32, 1e-4, "bertbase"), (torch.bfloat16, 4e-2, "bertbase")])
def test_pixel_head_decodes_an_image_and_a_reconstruction_scalar(golden_dir, dt, tol, case):
    """decode_image = forward -> to_pixels -> unpatchify, reconstruction_loss = the reference's own pixel terms (loss/mse.py:9-19,
    loss/discriminator.py:161,170) — against the oracle's restatement on the oracle's decoder output."""
    dc, sd, x, mask, _ = _case(golden_dir, case)
    det = SetokDeTokenizer(token_feat_dim=dc.token_feat_dim, hidden_dim=dc.hidden_dim, patch_size=dc.patch_size,
                           image_size=dc.image_size, decoder_embed_dim=dc.decoder_embed_dim, decoder_nheads=dc.decoder_nheads,
                           decoder_depth=dc.decoder_depth, mlp_ratio=dc.mlp_ratio,
                           feature_mapper_path_or_name=dict(hidden_size=dc.mapper_hidden, num_attention_heads=dc.mapper_heads,
                                                            intermediate_size=dc.mapper_intermediate, layer_norm_eps=dc.mapper_eps),
                           num_hidden_layers=dc.num_hidden_layers, cross_attention_freq=dc.cross_attention_freq, pixel_head=True)
    res = det.load_state_dict(sd, strict=False)
    assert set(res.missing_keys) <= {"position_embedding.inv_freq", "to_pixels.weight", "to_pixels.bias"} and not res.unexpected_keys
    g = torch.Generator().manual_seed(9)
    p = dc.patch_size
    with torch.no_grad():
        det.to_pixels.weight.copy_(torch.randn(det.to_pixels.weight.shape, generator=g) * 0.05)
        det.to_pixels.bias.copy_(torch.randn(det.to_pixels.bias.shape, generator=g) * 0.1)
    wpx, bpx = det.to_pixels.weight.detach().clone(), det.to_pixels.bias.detach().clone()
    det = det.to(device=DEV, dtype=dt).eval()
    B = x.shape[0]
    gh = dc.image_size // p
    img = det.decode_image(x.to(DEV), mask.to(DEV))
    assert img.shape == (B, 3, gh * p, gh * p) and img.dtype == dt
    feats = O.detokenizer_forward(sd, dc, x, mask)                                  # (B, Q, D) fp32 oracle
    want = O.unpatchify(feats.reshape(B * gh * gh, -1) @ wpx.t() + bpx, B, gh, gh, p)
    assert _rel(img.float(), want) < tol
    gold = torch.randn(want.shape, generator=g)
    for kind in ("mse", "l1"):
        got = det.reconstruction_loss(x.to(DEV), gold.to(DEV), mask.to(DEV), kind=kind)
        assert got.dim() == 0 and got.dtype == torch.float32
        ref = O.pixel_loss(want, gold if dt == torch.float32 else gold.bfloat16().float(), kind)
        assert abs(float(got) - float(ref)) < tol * abs(float(ref)), (kind, float(got), float(ref))
        again = det.reconstruction_loss(x.to(DEV), gold.to(DEV), mask.to(DEV), kind=kind)
        assert float(got) == float(again)                                            # fixed-order reduction
    with pytest.raises(ValueError, match="gold_image"):
        det.reconstruction_loss(x.to(DEV), gold[:, :, :-1].to(DEV), mask.to(DEV))
    # the rearrangement alone, bit for bit, on a padded row stride
    from setok_amd import ops
    pt = torch.randn(B * gh * gh, 3 * p * p + 5, generator=g).to(dt)
    assert torch.equal(ops.unpatchify(pt.to(DEV), B, gh, gh, p).cpu(), O.unpatchify(pt[:, : 3 * p * p], B, gh, gh, p))


def test_detokenizer_without_pixel_head_says_so(golden_dir):
    dc, sd, x, mask, _ = _case(golden_dir, "small")
    det = _build(dc, sd)
    assert det.to_pixels is None and "to_pixels.weight" not in det.state_dict()      # default: the reference's parameter tree, key for key
    with pytest.raises(RuntimeError, match="pixel_head=True"):
        det.decode_image(x.to(DEV), mask.to(DEV))
