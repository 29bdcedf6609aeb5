"""Full-size parity and property tests of the BASELINE.json configurations on a real MI355X (`pytest -m gpu`):

  cfg2  the bf16 throughput mode's CONTRACT at ViT-L/14-224 dims: from the GPU's own bf16 features the clustering decisions equal the
        fp32 oracle's wherever they are certain, token counts are equal, tokens agree within a stated bf16 tolerance
  cfg3  encode + reconstruction decoder at batch 256 (determinism, batch invariance)
  cfg4  ViT-L/14-336 (576 patches, T = 577): fp32 parity against the reference's own golden vectors (tests/golden/vitl_336.npz),
        bf16 batch invariance at full dims
  cfg5  Vicuna-7B layer dims (hidden 4096, 32 x 128 heads, SwiGLU 11008, vocab 32000): two layers against HuggingFace's golden
        outputs in fp32; all 32 layers, batch 32, as a determinism / batch-invariance / loss-vs-oracle property run in bf16

Everything goes through the C ABI (ctypes -> libsetok_hip.so); the oracle is the checker only."""
import os

import numpy as np
import pytest
import torch

import parity
import setok_oracle as O

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import setok_amd
    from setok_amd import SetokDeTokenizer, SetokTokenizer, ops
    from setok_amd.llama import SetokimLlamaPrefill
    from setok_amd.synthetic import init_synthetic_

DEV = "cuda"
TOL = 1e-4                # north_star: cluster feature tensors within 1e-4 relative (fp32 parity mode)
BF16_TOKEN_TOL = 1.0e-2   # bf16 throughput mode: tokens of the GPU head vs the fp32 oracle head on the SAME (bf16-valued) features and the
                          # same cluster assignment; max-abs error relative to the largest token entry.  MEASURED (profiles/r03_bf16_parity.txt):
                          # 5.1e-3 at cfg2 dims (8 images), 5.2e-3 at cfg4 dims; the reference's own bf16 run sits at 5.2-5.7e-3 on the same
                          # stage (tests/golden/bf16_reference.npz) — the tolerance is 2 x the measured value (round 2 asserted 4e-2)
# (the bf16 tower from pixels is held to 1.5 x the reference's own bf16 tower: tests/golden/bf16_tower.npz, measured 2.36e-2 max-rel / 1.24e-2 rms-rel)


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _rel(got, ref):
    got, ref = got.detach().double().cpu(), ref.double().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def _vitl_tok(img=224, dtype=torch.float32, threshold=0.125, with_tower=True, synthetic=False):
    vc = dict(vars(O.VitConfig(image_size=img)))
    if not with_tower:
        vc["num_hidden_layers"] = 0
    tok = SetokTokenizer(vision_tower=vc, hidden_dim=1024, token_feat_dim=4096, min_cluster_num=64, threshold=threshold,
                         nheads=2, dim_feedforward=4096)
    if synthetic:
        init_synthetic_(tok, 0, 1)
    else:
        sd = O.init_head_weights(O.HeadConfig(threshold=threshold), seed=1)
        if with_tower:
            sd.update(O.init_tower_weights(O.VitConfig(image_size=img), seed=0))
        assert not tok.load_state_dict(sd, strict=False).unexpected_keys
    return tok.to(device=DEV, dtype=dtype).eval()


def _oracle_head_from_x(sd, hc, x, labels):
    """tokenizer.py:177-180 on given features x (N, C) and a given assignment: group_encoding -> inter_encoder -> out."""
    group = O.group_encoding(sd, hc, x, labels)
    inter = O.block_forward(sd, "inter_encoder.", group, hc.nheads, hc.intra_cluster_layers)
    return torch.nn.functional.linear(inter, sd["out.weight"], sd["out.bias"])


def _own_features_contract(tok, st, idx, toks, grid, n_images, sd_head, hc, ulps=16.0):
    """The bf16 mode's contract: take the GPU's OWN bf16 features x (after the positional add), run the fp32 oracle's clustering on
    x.float() (bf16 values are exact in fp32) and demand (a) equality of every decision the fp64 margin analysis calls certain — the GPU
    computes d^2 = |a|^2 + |b|^2 - 2 a.b from exact bf16 products with fp32 accumulation, i.e. the reference's formula up to summation
    order — (b) equal token counts, (c) tokens within BF16_TOKEN_TOL of the fp32 oracle head run on the same x and the same assignment.
    Returns the statistics it asserted on."""
    N = grid * grid
    x = st["x"].float().cpu().reshape(-1, N, st["x"].shape[-1])
    out = dict(images=n_images, centres_certain=0, tokens_certain=0, tokens_equal=0, tokens_total=0, max_token_err=0.0)
    for i in range(n_images):
        ref = O.cluster_dpc_knn(x[i], hc.min_cluster_num, hc.threshold, hc.min_cluster_num)
        sens = O.cluster_sensitivity(x[i], hc.min_cluster_num, hc.threshold, hc.min_cluster_num, ulps=ulps)
        L = st["counts"][i]
        stats = O.check_cluster_parity(st["index_down"][i, :L].cpu(), idx[i].cpu(), ref.index_down, ref.idx_cluster, sens)   # raises on a certain mismatch
        out["centres_certain"] += stats["centres_certain"]
        out["tokens_certain"] += stats["tokens_certain"]
        out["tokens_total"] += N
        if stats["centres_certain"]:
            assert L == ref.index_down.numel()                                                  # per-image token count
            out["tokens_equal"] += stats["tokens_equal"]
        want = _oracle_head_from_x(sd_head, hc, x[i], idx[i].cpu())                             # same assignment: isolates the arithmetic
        err = _rel(toks[i].float(), want)
        out["max_token_err"] = max(out["max_token_err"], err)
        assert toks[i].shape == want.shape and err < BF16_TOKEN_TOL, (i, err)
    return out


# ======================================================================================================================================
# the bf16 mode against the REFERENCE'S OWN bf16 run (tests/golden/bf16_reference.npz: the reference's head cast to torch.bfloat16, as
# train_setokim.py:326 casts the module, run on CPU from fixed bf16 features)
# ======================================================================================================================================
BF16_VS_REFERENCE = 1.5    # a GPU stage may be at most this factor further from the fp32 oracle than the reference's own bf16 arithmetic is


LOW = {"bf16": torch.bfloat16, "fp16": torch.float16}      # the two 16-bit modes: bf16 = the BASELINE metric's; fp16 = what the reference's inference loader and its
                                                            # non-`--bf16` launches cast the tower to (src/model/builder.py:43,135-136, train_setokim.py:326) — round 6
FP16_TOKEN_TOL = 2.0e-3    # fp16 mode: tokens vs the fp32 oracle head on the same fp16-valued features (11-bit significands: ~8 x tighter than bf16's 1.0e-2)


@pytest.mark.parametrize("low", ["bf16", "fp16"])
@pytest.mark.parametrize("cfg,src,grid", [("cfg2", "vitl_224", 16), ("cfg4", "vitl_336", 24)])
def test_bf16_mode_is_no_worse_than_the_references_own_bf16_run(golden_dir, cfg, src, grid, low):
    """(low = "fp16": the same statement for the fp16 mode against tests/golden/fp16_reference.npz — the reference's head cast to torch.float16.)
    Per stage (group, inter, tokens): |GPU bf16 - fp32 oracle| <= 1.5 x |reference bf16 - fp32 oracle|, each measured on ITS OWN bf16-valued
    features x and ITS OWN cluster assignment (the reference's bf16 clustering rounds the scores to bf16 and picks a slightly different L than the
    fp32 clustering of the same features, which is what the GPU computes: 37 / 46 against 36 / 47 at cfg2), with the bf16-rounded weights.
    The measured errors are printed (-s) and recorded in DESIGN.md §2."""
    dt = LOW[low]
    z = np.load(os.path.join(golden_dir, low + "_reference.npz"))
    feats = _t(np.load(os.path.join(golden_dir, src + ".npz"))["feats"])
    hc = O.HeadConfig(threshold=0.125)
    sd_head = {k: v.to(dt).float() for k, v in O.init_head_weights(hc, seed=1).items()}
    tok = _vitl_tok(img=14 * grid, dtype=dt, with_tower=False)
    B, N = feats.shape[0], grid * grid
    hidden = torch.cat([torch.cat([torch.zeros(1, 1024), f], 0) for f in feats], 0).to(DEV, dt)      # class-token rows that 'patch' drops
    toks, idx, score, st = tok.encode_features(hidden, B, return_stages=True)
    x = st["x"].float().cpu().reshape(B, N, -1)
    offs = np.concatenate([[0], np.cumsum(st["counts"])])
    worst = dict(group=(0.0, 0.0), inter=(0.0, 0.0), tokens=(0.0, 0.0))
    for i in range(B):
        lab = idx[i].cpu()
        group = O.group_encoding(sd_head, hc, x[i], lab)
        inter = O.block_forward(sd_head, "inter_encoder.", group, hc.nheads, hc.intra_cluster_layers)
        tokens = torch.nn.functional.linear(inter, sd_head["out.weight"], sd_head["out.bias"])
        mine = dict(group=_rel(st["group"][offs[i]: offs[i + 1]].float(), group), inter=_rel(st["inter"][offs[i]: offs[i + 1]].float(), inter),
                    tokens=_rel(toks[i].float(), tokens))
        ref = dict(zip(("group", "inter", "tokens"), z[f"{cfg}:{i}:errs"].tolist()))
        f32_L = int(z[f"{cfg}:{i}:L"][1])
        # (which decisions must be EQUAL to the fp32 clustering is the contract test's business below: certain ones; here the counts are reported)
        assert abs(st["counts"][i] - f32_L) <= max(2, f32_L // 50), (st["counts"][i], f32_L)
        assert mine["tokens"] < (BF16_TOKEN_TOL if low == "bf16" else FP16_TOKEN_TOL)
        print(f"{cfg} image {i}: L = {st['counts'][i]} (fp32 clustering of the same features: {f32_L}, reference {low}: {int(z[f'{cfg}:{i}:L'][0])});  " +
              "  ".join(f"{k}: gpu {mine[k]:.3e} / reference-{low} {ref[k]:.3e}" for k in mine))
        for k in mine:
            assert mine[k] <= BF16_VS_REFERENCE * ref[k], (cfg, i, k, mine[k], ref[k])
            if mine[k] / ref[k] > worst[k][0] / max(worst[k][1], 1e-30):
                worst[k] = (mine[k], ref[k])
    print(f"{cfg} {low} worst ratios:", {k: round(a / b, 3) for k, (a, b) in worst.items()})


# ======================================================================================================================================
# cfg2 — the bf16 mode's contract at ViT-L/14-224 dims
# ======================================================================================================================================
def test_cfg2_bf16_contract_on_the_gpus_own_features():
    hc = O.HeadConfig(threshold=0.125)
    sd_head = O.init_head_weights(hc, seed=1)
    sd_head = {k: v.bfloat16().float() for k, v in sd_head.items()}          # the GPU head holds bf16 weights: compare like with like
    tok = _vitl_tok(dtype=torch.bfloat16)
    g = torch.Generator().manual_seed(3)
    B = 8
    images = torch.randn(B, 3, 224, 224, generator=g)
    hidden = tok.image_feature_encoder.hidden_rows(images.to(DEV))
    toks, idx, score, st = tok.encode_features(hidden, B, return_stages=True)
    s = _own_features_contract(tok, st, idx, toks, 16, B, sd_head, hc)
    print("cfg2 bf16 contract:", s)
    # floors: on these seeded inputs (almost) every decision is certain; a vacuous pass must be impossible
    assert s["centres_certain"] >= B - 1
    assert s["tokens_certain"] >= 0.97 * s["tokens_total"] * s["centres_certain"] / B
    assert s["tokens_equal"] >= s["tokens_certain"]


def _drift(index_down_a, idx_a, index_down_b, idx_b):
    """(|L_a - L_b|, 1 - Jaccard of the centre-token sets, fraction of tokens whose centre TOKEN differs) — as tests/golden/make_golden.py's
    partition_drift measures the reference's own bf16 run against its fp32 run."""
    a, b = set(index_down_a.tolist()), set(index_down_b.tolist())
    return abs(len(a) - len(b)), 1.0 - len(a & b) / float(len(a | b)), 1.0 - float((index_down_a[idx_a] == index_down_b[idx_b]).float().mean())


@pytest.mark.parametrize("low", ["bf16", "fp16"])
@pytest.mark.parametrize("sel", [-2, -1])
def test_cfg2_bf16_from_pixels_drifts_no_further_than_the_references_own_bf16_run(golden_dir, sel, low):
    """(low = "fp16": `tok.to(dtype=torch.float16)` then `tok(images)` — what src/model/builder.py:135-136 does to the tower — against
    tests/golden/fp16_tower.npz, the reference's tower + head cast to torch.float16 and run on CPU.)
    Throughput mode (bf16 end to end) FROM PIXELS against the reference's fp32 run on the same seeded weights / images.
    The yardstick is the REFERENCE'S OWN bf16 run from pixels (tests/golden/bf16_tower.npz: HF CLIP tower + the reference head cast to
    torch.bfloat16 on CPU, train_setokim.py:326):
      * tower: |GPU bf16 features - reference fp32 features| <= 1.5 x |reference bf16 features - reference fp32 features| (max-rel and rms-rel);
        this is the gemm_pp + LayerNorm-fold + attn_vit chain that is 95 % of the timed step;
      * clustering (discontinuous in its input, so integers cannot be demanded from a 2e-2-perturbed input): token count, centre set and
        partition drift from the fp32 run <= 1.5 x the largest drift the reference's bf16 run shows on these images (round 3 asserted loose
        floors: counts within 20 %, Jaccard >= 0.5, same-centre >= 0.4).
    sel = -2: the reference classes' default layer; sel = -1: what the launch scripts pass (bench.py's headline)."""
    dt = LOW[low]
    zb = np.load(os.path.join(golden_dir, "bf16_tower.npz"))                     # (holds the reference's fp32 run at select_layer = -1 for both modes)
    zt = zb if low == "bf16" else np.load(os.path.join(golden_dir, "fp16_tower.npz"))
    if sel == -2:
        z = np.load(os.path.join(golden_dir, "vitl_224.npz"))
        feats32 = _t(z["feats"])
        ref32 = [(_t(z[f"{i}:index_down"]).long(), _t(z[f"{i}:idx_cluster"]).long()) for i in range(2)]
    else:
        feats32 = _t(zb["vitl:sel-1:feats32"])
        ref32 = [(_t(zb[f"vitl:sel-1:{i}:index_down32"]).long(), _t(zb[f"vitl:sel-1:{i}:idx_cluster32"]).long()) for i in range(2)]
    tag = f"vitl:sel{sel}"
    ref_max, ref_rms = zt[tag + ":tower_err"].tolist()
    ref_drift = np.stack([zt[f"{tag}:{i}:drift"] for i in range(2)]).max(axis=0)          # the reference-bf16 run's worst image, per measure
    images = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(3))
    tok = _vitl_tok(dtype=dt)
    tok.image_feature_encoder.select_layer = sel
    hidden = tok.image_feature_encoder.hidden_rows(images.to(DEV))
    toks, idx, score, st = tok.encode_features(hidden, 2, return_stages=True)
    feats = tok.image_feature_encoder(images.to(DEV).to(dt))
    assert feats.dtype == dt
    feats = feats.float().cpu()
    rms = lambda a, b: float(((a.double() - b.double()) ** 2).mean().sqrt() / (b.double() ** 2).mean().sqrt())
    got_max, got_rms = _rel(feats, feats32), rms(feats, feats32)
    both = f"; GPU vs reference-bf16 rms-rel {rms(feats, torch.from_numpy(zt[tag + ':feats_bf16_bits']).view(torch.bfloat16).float()):.3e}" if low == "bf16" else ""
    print(f"{low} tower from pixels (select_layer {sel}): GPU vs reference-fp32 max-rel {got_max:.3e} rms-rel {got_rms:.3e}; "
          f"reference-{low} vs reference-fp32 max-rel {ref_max:.3e} rms-rel {ref_rms:.3e}" + both)
    assert got_max <= BF16_VS_REFERENCE * ref_max and got_rms <= BF16_VS_REFERENCE * ref_rms
    for i in range(2):
        L = st["counts"][i]
        dl, dj, dp = _drift(st["index_down"][i, :L].cpu(), idx[i].cpu(), *ref32[i])
        print(f"  image {i}: L = {L} (reference fp32 {ref32[i][0].numel()}, reference {low} {int(zt[f'{tag}:{i}:L'][0])}); 1 - centre Jaccard {dj:.3f} "
              f"(reference-{low} worst {ref_drift[1]:.3f}); tokens with another centre {dp:.3f} (reference-{low} worst {ref_drift[2]:.3f})")
        # (fp16 sits 8 x closer to fp32 than bf16: where the reference's fp16 run does not drift at all on these two images, one boundary token of
        #  slack stands in for "1.5 x 0")
        slack = (0, 0.0, 0.0) if low == "bf16" else (1, 0.05, 0.02)
        assert dl <= BF16_VS_REFERENCE * ref_drift[0] + slack[0] and dj <= BF16_VS_REFERENCE * ref_drift[1] + slack[1] and \
            dp <= BF16_VS_REFERENCE * ref_drift[2] + slack[2], (i, dl, dj, dp, ref_drift)
    assert all(t.shape[1] == 4096 and torch.isfinite(t.float()).all() for t in toks)


@pytest.mark.parametrize("low", ["bf16", "fp16"])
def test_small_dims_bf16_tower_from_pixels_against_the_references_bf16_tower(golden_dir, low):
    """The same yardstick at small dims with the weights in the fixture (the whole tower, 4 images): GPU bf16 (fp16) features no further from the
    reference's fp32 features than 1.5 x the reference's own bf16 (fp16) tower."""
    dt = LOW[low]
    zt = np.load(os.path.join(golden_dir, low + "_tower.npz"))
    sd = {k[len("small:w:"):]: _t(zt[k]) for k in zt.files if k.startswith("small:w:")}
    vc = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=4, image_size=112, patch_size=14)
    tok = SetokTokenizer(vision_tower=vc, mm_vision_select_layer=-2, hidden_dim=64, token_feat_dim=96, min_cluster_num=8, threshold=0.5, nheads=2,
                         dim_feedforward=128)
    assert not tok.load_state_dict(sd, strict=False).unexpected_keys
    tok = tok.to(device=DEV, dtype=dt).eval()
    images = torch.randn(4, 3, 112, 112, generator=torch.Generator().manual_seed(21))
    feats = tok.image_feature_encoder(images.to(DEV).to(dt)).float().cpu()
    feats32 = _t(zt["small:feats32"])
    rms = lambda a, b: float(((a.double() - b.double()) ** 2).mean().sqrt() / (b.double() ** 2).mean().sqrt())
    ref_max, ref_rms = zt["small:tower_err"].tolist()
    got_max, got_rms = _rel(feats, feats32), rms(feats, feats32)
    print(f"small dims {low} tower: GPU max-rel {got_max:.3e} rms-rel {got_rms:.3e}; reference-{low} max-rel {ref_max:.3e} rms-rel {ref_rms:.3e}")
    assert got_max <= BF16_VS_REFERENCE * ref_max and got_rms <= BF16_VS_REFERENCE * ref_rms


# ======================================================================================================================================
# cfg4 — ViT-L/14-336: 576 patches, T = 577
# ======================================================================================================================================
def test_cfg4_vitl336_head_fp32_from_reference_features(golden_dir):
    """The head at cfg4 dims on the reference's own fp32 tower features: integers bit-exact, tensors within 1e-4."""
    z = np.load(os.path.join(golden_dir, "vitl_336.npz"))
    tok = _vitl_tok(img=336, with_tower=False)
    feats = _t(z["feats"])
    assert tuple(feats.shape) == (2, 576, 1024)
    hidden = torch.cat([torch.zeros(2, 1, 1024), feats], 1).reshape(-1, 1024).to(DEV)
    toks, idx, score, st = tok.encode_features(hidden, 2, return_stages=True)
    x = feats + O.pos_encoding_2d(24, 24, 1024)[None]
    start = 0
    for i in range(2):
        sens = O.cluster_sensitivity(x[i], 64, 0.125, 64)
        L = st["counts"][i]
        stats = O.check_cluster_parity(st["index_down"][i, :L].cpu(), idx[i].cpu(), _t(z[f"{i}:index_down"]).long(),
                                       _t(z[f"{i}:idx_cluster"]).long(), sens)
        assert sens["centres_certain"] and stats["tokens_equal"] == 576          # bit-exact indices and token count (fixture: all certain)
        assert L == _t(z[f"{i}:index_down"]).numel()
        O.check_score(score[i].cpu(), sens)
        parity.close(st["group"][start:start + L], _t(z[f"{i}:group"]), TOL, "st['group'][start:start + L], _t(z[f'{i}:group'])")
        parity.close(toks[i], _t(z[f"{i}:tokens"]), TOL, "toks[i], _t(z[f'{i}:tokens'])")
        start += L


def test_cfg4_vitl336_tower_fp32_and_whole_path_from_pixels(golden_dir):
    """a1 at ViT-L/14-336 in fp32 (exact-f32 MFMA GEMMs, the 19-tile attention path) against the reference's HF tower, then the whole
    path from pixels: every decision the margin analysis calls certain equals the reference's."""
    z = np.load(os.path.join(golden_dir, "vitl_336.npz"))
    tok = _vitl_tok(img=336)
    g = torch.Generator().manual_seed(int(z["spec"][2]))
    images = torch.randn(2, 3, 336, 336, generator=g)
    feats = tok.image_feature_encoder(images.to(DEV))
    assert tuple(feats.shape) == (2, 576, 1024)
    ferr = _rel(feats, _t(z["feats"]))
    assert ferr < TOL
    toks, idx, score = tok(images.to(DEV))
    x = _t(z["feats"]) + O.pos_encoding_2d(24, 24, 1024)[None]
    certain = compared = n_diff = 0
    for i in range(2):
        # the tower features differ from the reference's by ~1e-5 relative (fp32 summation order through 23 layers), i.e. ~100 ulps of
        # |x|^2 in d^2: widen the perturbation accordingly
        sens = O.cluster_sensitivity(x[i], 64, 0.125, 64, ulps=256.0)
        same = idx[i].cpu() == _t(z[f"{i}:idx_cluster"]).long()
        n_diff += int((~same).sum())
        certain += int(sens["centres_certain"])
        if sens["centres_certain"]:
            assert toks[i].shape[0] == _t(z[f"{i}:index_down"]).numel()
            assert bool((same | ~sens["assign_certain"]).all())
            compared += int(sens["assign_certain"].sum())
            if bool(same.all()):
                parity.close(toks[i], _t(z[f"{i}:tokens"]), TOL, "toks[i], _t(z[f'{i}:tokens'])")
    print(f"cfg4 fp32 from pixels: tower rel err {ferr:.2e}; images with certain centres {certain}/2; certain assignments compared {compared}/1152; "
          f"tokens with a different cluster id {n_diff}")
    assert certain == 2 and compared >= 0.97 * 1152                              # the fixture's decisions are certain: no vacuous pass


def test_cfg4_full_dims_bf16_batch_invariance_and_contract():
    """cfg4 at full dims in bf16, one GPU's share of the batch (16 images of 336^2): determinism, bit-exact batch invariance, the
    structural invariants of the clustering at N = 576, and the bf16 contract on the GPU's own features for two of the images."""
    tok = _vitl_tok(img=336, dtype=torch.bfloat16, synthetic=True)
    g = torch.Generator().manual_seed(13)
    images = torch.randn(16, 3, 336, 336, generator=g).to(DEV, torch.bfloat16)
    toks, idx, score = tok(images)
    toks2, idx2, score2 = tok(images)
    assert torch.equal(toks.packed, toks2.packed) and torch.equal(idx, idx2) and torch.equal(score, score2)
    counts = torch.tensor(toks.counts)
    assert tuple(idx.shape) == (16, 576) and int(counts.sum()) == toks.packed.shape[0] and toks.packed.shape[1] == 4096
    assert int(counts.min()) >= 1 and int(counts.max()) <= 576 and bool(torch.isfinite(toks.packed.float()).all())
    assert torch.equal(idx.max(dim=1).values.cpu() + 1, counts)
    pick = [1, 7, 15]
    sub_t, sub_i, sub_s = tok(images[pick])
    for j, i in enumerate(pick):
        assert torch.equal(sub_i[j], idx[i]) and torch.equal(sub_s[j], score[i]) and torch.equal(sub_t[j], toks[i])
    hc = O.HeadConfig(threshold=0.125)
    sd_head = {k: v.detach().float().cpu() for k, v in tok.state_dict().items() if k.split(".")[0] in ("inner_encoder", "inter_encoder", "out")}
    hidden = tok.image_feature_encoder.hidden_rows(images[:2])
    t2, i2, s2, st = tok.encode_features(hidden, 2, return_stages=True)
    s = _own_features_contract(tok, st, i2, t2, 24, 2, sd_head, hc)
    print("cfg4 bf16 contract:", s)
    assert torch.equal(i2, idx[:2])


# ======================================================================================================================================
# cfg3 — encode + reconstruction decoder at batch 256
# ======================================================================================================================================
def test_cfg3_decoder_full_dims_batch_256():
    """cfg3 at its batch size: 256 images' tokens (24-55 each) through the reconstruction decoder at full dims in bf16: two runs are
    bit-identical and an image decoded in a small batch is bit-identical to the same image inside the batch of 256."""
    det = SetokDeTokenizer(token_feat_dim=4096, hidden_dim=768, patch_size=14, image_size=256, decoder_embed_dim=768, decoder_nheads=16,
                           decoder_depth=16).to(device=DEV, dtype=torch.bfloat16).eval()
    g = torch.Generator().manual_seed(6)
    counts = [int(c) for c in torch.randint(24, 56, (256,), generator=g)]
    packed = torch.randn(sum(counts), 4096, generator=g).to(device=DEV, dtype=torch.bfloat16)
    toks = list(packed.split(counts))
    full = det(toks)
    assert tuple(full.shape) == (256, 324, 768) and bool(torch.isfinite(full.float()).all())
    assert torch.equal(full, det(toks))
    sub = det([toks[i] for i in (3, 130, 255)])
    for j, i in enumerate((3, 130, 255)):
        assert torch.equal(sub[j], full[i]), i


# ======================================================================================================================================
# cfg5 — Vicuna-7B layer dims
# ======================================================================================================================================
def test_cfg5_vicuna7b_dims_two_layers_fp32_vs_hf(golden_dir):
    """hidden 4096 / 32 heads x 128 / SwiGLU 11008 / vocab 32000, two decoder layers, fp32: hidden states and logits against HuggingFace
    LlamaForCausalLM's golden outputs (generated by tests/golden/make_golden.py; the 0.67 G weights regenerate from the seed)."""
    z = np.load(os.path.join(golden_dir, "llama_7bdims.npz"))
    kw = {str(k): int(v) for k, v in zip(z["cfg_keys"], z["cfg_vals"])}
    seed, B, T, left = (int(v) for v in z["spec"])
    lc = O.LlamaConfigLite(**kw)
    assert (lc.hidden_size, lc.intermediate_size, lc.num_attention_heads, lc.head_dim, lc.vocab_size) == (4096, 11008, 32, 128, 32000)
    sd = O.init_llama_weights(lc, seed=seed)
    x, am, pos = O.llama_inputs(lc, seed, B, T, "left" if left else "right")
    with torch.device("meta"):
        m = SetokimLlamaPrefill(kw)
    m = m.to_empty(device=DEV)
    assert not m.load_state_dict(sd, strict=True).missing_keys
    m.eval()
    del sd
    hidden = m.model(x.to(DEV), am.to(DEV), pos.to(DEV))
    v = am.bool()
    parity.close(hidden.cpu()[v], _t(z["hidden"])[v], TOL, "hidden.cpu()[v], _t(z['hidden'])[v]")
    lg, _, _ = m(inputs_embeds=x.to(DEV), attention_mask=am.to(DEV), position_ids=pos.to(DEV))
    assert tuple(lg.shape) == (B, T, 32000)
    stride = 32000 // _t(z["logits_cols"]).shape[-1] + 1
    parity.close(lg.cpu()[:, :, ::stride][v], _t(z["logits_cols"])[v], TOL, "lg.cpu()[:, :, ::stride][v], _t(z['logits_cols'])[v]")
    last = _t(z["last"]).tolist()
    got_last = torch.stack([lg[b, t] for b, t in enumerate(last)]).cpu()
    parity.close(got_last, _t(z["logits_last"]), TOL, "got_last, _t(z['logits_last'])")
    only_last, _, _ = m(inputs_embeds=x.to(DEV), attention_mask=am.to(DEV), position_ids=pos.to(DEV), last_token_only=True)
    parity.close(only_last.cpu(), _t(z["logits_last"]), TOL, "only_last.cpu(), _t(z['logits_last'])")


def test_cfg5_full_depth_batch32_properties_bf16():
    """The full Setokim forward of cfg5 — 32 images -> SeTok encode -> projector -> splice into 512-token prompts -> 32-layer LLM at
    Vicuna-7B dims -> logits -> language-model loss — in bf16 with seeded random weights: two runs are bit-identical, a sub-batch's
    sequences are bit-identical to the same sequences inside the batch of 32 (valid positions), and the loss kernel agrees with the oracle's
    restatement of setokim_llama.py:145-160 on the same logits."""
    tok = _vitl_tok(dtype=torch.bfloat16, synthetic=True)
    proj = setok_amd.build_vision_projector("mlp2x_gelu", mm_hidden_size=4096, hidden_size=4096)
    torch.manual_seed(2)
    for mod in proj:
        if isinstance(mod, torch.nn.Linear):
            torch.nn.init.xavier_uniform_(mod.weight); torch.nn.init.zeros_(mod.bias)
    proj = proj.to(device=DEV, dtype=torch.bfloat16).eval()
    lcfg = dict(vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
                num_key_value_heads=32, rms_norm_eps=1e-5, rope_theta=10000.0)
    with torch.device(DEV):
        llm = SetokimLlamaPrefill(lcfg, vision_tower=tok, mm_in_projector=proj).to(torch.bfloat16)
    gl = torch.Generator(device=DEV).manual_seed(11)
    for n_, p_ in llm.named_parameters():
        if n_.startswith(("vision_tower.", "mm_in_projector.")):
            continue
        if p_.dim() == 2:
            p_.data.normal_(0.0, 0.02, generator=gl)
        else:
            p_.data.fill_(1.0)
    llm.eval()
    B, T = 32, 512
    g = torch.Generator().manual_seed(5)
    images = torch.randn(B, 3, 224, 224, generator=g).to(DEV, torch.bfloat16)
    ids = torch.randint(0, 32000, (B, T), generator=g)
    ids[:, 17] = -200
    am = torch.ones(B, T, dtype=torch.bool)
    for b in range(1, B, 3):
        am[b, T - 40 - b:] = False                                     # ragged right padding
    labels = ids.clone(); labels[:, :64] = -100
    ids, am, labels = ids.to(DEV), am.to(DEV), labels.to(DEV)
    logits, new_labels, new_am, loss = llm(input_ids=ids, attention_mask=am, labels=labels, comp_images=images, return_loss=True)
    counts = list(llm._last_features.counts)
    assert logits.shape[0] == B and logits.shape[2] == 32000 and logits.shape[1] == max(int(am[b].sum()) - 1 + counts[b] for b in range(B))
    assert bool(torch.isfinite(loss)) and bool(torch.isfinite(logits[new_am.bool()].float()).all())
    logits2, _, _, loss2 = llm(input_ids=ids, attention_mask=am, labels=labels, comp_images=images, return_loss=True)
    assert torch.equal(logits, logits2) and torch.equal(loss, loss2)                                    # deterministic
    del logits2
    pick = [0, 4, 31]
    sub, sub_labels, sub_am, _ = llm(input_ids=ids[pick], attention_mask=am[pick], labels=labels[pick], comp_images=images[pick], return_loss=True)
    for j, b in enumerate(pick):
        n = int(sub_am[j].sum())
        assert n == int(new_am[b].sum()) and torch.equal(sub_labels[j, :n], new_labels[b, :n])
        assert torch.equal(sub[j, :n], logits[b, :n]), b                                                # batch-invariant, bit for bit
    # the loss on the same logits: kernel vs the oracle (CPU), on a slice that keeps the CPU leg short
    sl = slice(0, 6)
    want = float(O.lm_loss(logits[sl].float().cpu(), new_labels[sl].cpu(), new_am[sl].cpu()))
    got = float(ops.lm_loss(logits[sl].contiguous(), new_labels[sl].contiguous(), new_am[sl].contiguous())[0])
    assert abs(got - want) <= 2e-5 * abs(want), (got, want)
    print(f"cfg5 full depth: loss {float(loss):.4f}; slice loss kernel {got:.6f} vs oracle {want:.6f}; tokens/img {sum(counts) / B:.1f}")
