"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE (RAC, oracle/rac_harness.py).

Runs only in the build container (needs /root/reference).  The .npz files hold data only: inputs
(or the seeds that regenerate them bit-exactly with torch's CPU generator), weights for the small
cases, and the reference's outputs at every stage.  Usage:  python tests/golden/make_golden.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import rac_harness as R          # noqa: E402
import setok_oracle as O         # noqa: E402

torch.set_num_threads(8)


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"wrote {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def load_weights_into(tok, sd):
    missing, unexpected = tok.load_state_dict(sd, strict=False)
    bad = [m for m in missing if "layers." not in m or ".0." not in m.split("layers.")[1][:6]]
    # `layers.{i}.0.*` are aliases of norm1 (module.py:87-88): fine if "missing"
    assert not unexpected, unexpected
    assert all(".0.weight" in m or ".0.bias" in m or m.endswith("inv_freq") for m in missing), missing


def mid_threshold(tok, feats, rank, k=None, token_mask=None, noise=None):
    """A threshold halfway between the rank-th and (rank+1)-th largest reference scores, so that
    the dynamic-k branch yields L == rank with the widest possible decision margin."""
    st = R.rac_head_single(tok, feats, k=k, threshold=1e9, token_mask=token_mask, noise=noise, return_stages=True)
    s = torch.sort(st["score"].reshape(-1), descending=True).values
    return float((s[rank - 1] + s[rank]) / 2)


def small_tok(hidden=64, heads=4, layers=3, mlp=128, img=112, patch=14, ff=128, tfd=96, mcn=8, thr=0.5, sel=-2):
    d = R.make_clip_dir(hidden, layers, heads, mlp, img, patch, seed=0)
    return R.build_reference_tokenizer(d, hidden_dim=hidden, token_feat_dim=tfd, dim_feedforward=ff,
                                       min_cluster_num=mcn, threshold=thr, select_layer=sel)


# ---------------------------------------------------------------------------------------------
def gen_head_small():
    """Head (a2..a7) at small dims: all weights + inputs + every stage output."""
    tok = small_tok()
    sd = {k: v for k, v in tok.state_dict().items() if not k.startswith("image_feature_encoder")}
    cfg = dict(hidden_dim=64, token_feat_dim=96, nheads=2, dim_feedforward=128, min_cluster_num=8, threshold=0.5)
    g = torch.Generator().manual_seed(11)
    cases = {}
    N, C = 64, 64
    feats_rand = torch.randn(N, C, generator=g)
    noise = torch.rand(N, generator=g)
    mask = (torch.rand(N, generator=g) > 0.25).float()
    planted = O.planted_features(N, C, 5, seed=7)
    thr_dyn = mid_threshold(tok, feats_rand, 13, noise=noise)
    thr_msk = mid_threshold(tok, feats_rand, 9, token_mask=mask, noise=noise)
    specs = {
        "fallback":   dict(feats=feats_rand, k=None, threshold=None, token_mask=None, noise=noise),
        "dynamic":    dict(feats=feats_rand, k=None, threshold=thr_dyn, token_mask=None, noise=noise),
        "planted":    dict(feats=planted, k=6, threshold=None, token_mask=None, noise=None),
        "masked":     dict(feats=feats_rand, k=None, threshold=thr_msk, token_mask=mask, noise=noise),
        "k_explicit": dict(feats=feats_rand, k=3, threshold=0.9, token_mask=None, noise=None),
        "n16_direct": dict(feats=torch.randn(16, C, generator=g), k=4, threshold=None, token_mask=None, noise=None),
    }
    out = {"cfg_keys": np.array(list(cfg.keys())), "cfg_vals": np.array(list(cfg.values()), dtype=np.float64)}
    for k, v in sd.items():
        out["w:" + k] = npy(v)
    for name, s in specs.items():
        st = R.rac_head_single(tok, s["feats"], k=s["k"], threshold=s["threshold"], token_mask=s["token_mask"],
                               noise=s["noise"], return_stages=True)
        print(f"  head_small/{name}: L={st['tokens'].shape[0]}")
        out[f"{name}:feats"] = npy(s["feats"])
        out[f"{name}:k"] = np.array(-1 if s["k"] is None else s["k"])
        out[f"{name}:threshold"] = np.array(-1.0 if s["threshold"] is None else s["threshold"])
        if s["token_mask"] is not None:
            out[f"{name}:token_mask"] = npy(s["token_mask"])
        if s["noise"] is not None:
            out[f"{name}:noise"] = npy(s["noise"])
        for key in ("x", "index_down", "idx_cluster", "score", "group", "inter", "tokens"):
            out[f"{name}:{key}"] = npy(st[key])
    save("head_small", **out)


def gen_cluster_full():
    """cluster_dpc_knn (a3) at the BASELINE dims (C=1024; N=256 and 576) on seeded planted feature
    maps: only seeds + the reference's integer/score outputs are stored (inputs regenerate
    bit-exactly from oracle.planted_features)."""
    tok = small_tok(hidden=64)      # cluster_dpc_knn uses no weights; only min_cluster_num matters
    out = {}
    C = 1024
    for N in (256, 576):
        for m, k, thr, mcn in ((2, 8, 0.5, 64), (4, 8, 0.5, 64), (8, 8, 0.5, 64), (16, 8, 0.5, 64),
                               (12, 64, 0.5, 64), (8, 8, 1e9, 32)):
            tok.min_cluster_num = mcn
            x = O.planted_features(N, C, m, seed=100 + m) + O.pos_encoding_2d(int(N ** .5), int(N ** .5), C)
            with R.FixedNoise(None):
                idx_down, idx_cluster, score = tok.cluster_dpc_knn(x, k, None, thr)
            name = f"N{N}_m{m}_k{k}_mcn{mcn}_thr{thr:g}"
            print(f"  cluster_full/{name}: L={idx_down.numel()}")
            out[name + ":index_down"] = npy(idx_down).astype(np.int32)
            out[name + ":idx_cluster"] = npy(idx_cluster).astype(np.int32)
            out[name + ":score"] = npy(score)
            out[name + ":spec"] = np.array([N, C, m, 100 + m, k, mcn, thr], dtype=np.float64)
    save("cluster_full", **out)


def gen_e2e_small():
    """Tower + head end to end at small dims with all weights stored (ViT hidden 64, 3 layers,
    4 heads, 112^2 / patch 14 -> N=64)."""
    tok = small_tok(thr=0.5, mcn=8)
    sd = O.normalise_tower_keys(dict(tok.state_dict()))
    g = torch.Generator().manual_seed(3)
    images = torch.randn(3, 3, 112, 112, generator=g)
    noise = torch.rand(3, 64, generator=g)
    out = {}
    for k, v in sd.items():
        if ".layers." in k and k.split(".")[0] in ("inner_encoder", "inter_encoder") and k.split(".")[3] == "0":
            continue                      # aliases of norm1
        out["w:" + k] = npy(v)
    out["images"] = npy(images)
    out["noise"] = npy(noise)
    for sel in (-2, -1):
        tok.image_feature_encoder.select_layer = sel
        f0 = tok.image_feature_encoder(images)
        for thr in (0.5, round(mid_threshold(tok, f0[0], 11, noise=noise[0]), 4)):
            feats, res = R.rac_forward(tok, images, threshold=thr, noise=noise, return_stages=True)
            tag = f"sel{sel}_{'fallback' if thr == 0.5 else 'dynamic'}"
            out[f"{tag}:threshold"] = np.array(thr)
            out[f"{tag}:feats"] = npy(feats)
            for i, r in enumerate(res):
                print(f"  e2e_small/{tag}/img{i}: L={r['tokens'].shape[0]}")
                for key in ("index_down", "idx_cluster", "score", "tokens"):
                    out[f"{tag}:{i}:{key}"] = npy(r[key])
    save("e2e_small", **out)


def gen_tower_features_vitl():
    """Reference tower + head at the full ViT-L/14-224 dims (cfg2): weights from seeds
    (oracle.init_tower_weights / init_head_weights), 2 images.  Stores the reference's tower
    features in full (so clustering parity can start from IDENTICAL fp32 features)
    and every downstream output."""
    from transformers import CLIPVisionConfig, CLIPVisionModel
    vc = O.VitConfig()                                   # ViT-L/14-224
    hc = O.HeadConfig(threshold=0.125)                   # dyn-k fires with random-init features (§8d)
    tsd = O.init_tower_weights(vc, seed=0)
    hsd = O.init_head_weights(hc, seed=1)
    d = R.make_clip_dir(vc.hidden_size, vc.num_hidden_layers, vc.num_attention_heads, vc.intermediate_size,
                        vc.image_size, vc.patch_size, seed=0)
    tok = R.build_reference_tokenizer(d, hidden_dim=1024, token_feat_dim=4096, dim_feedforward=4096,
                                      min_cluster_num=64, threshold=0.125, select_layer=-2)
    full = dict(tsd); full.update(hsd)
    load_weights_into(tok, full)
    g = torch.Generator().manual_seed(3)
    images = torch.randn(2, 3, 224, 224, generator=g)
    out = {"spec": np.array([0, 1, 3], dtype=np.int64)}   # tower seed, head seed, image seed
    feats, res = R.rac_forward(tok, images, noise=None, return_stages=True)
    out["feats"] = npy(feats)
    for i, r in enumerate(res):
        print(f"  vitl/img{i}: L={r['tokens'].shape[0]} score[{r['score'].min():.4f},{r['score'].max():.4f}]")
        out[f"{i}:index_down"] = npy(r["index_down"]).astype(np.int32)
        out[f"{i}:idx_cluster"] = npy(r["idx_cluster"]).astype(np.int32)
        out[f"{i}:score"] = npy(r["score"])
        out[f"{i}:group"] = npy(r["group"]).astype(np.float32)
        out[f"{i}:tokens"] = npy(r["tokens"]).astype(np.float32)
    # fallback branch on the same features (default threshold 0.5 -> topk 64)
    for i in range(2):
        r = R.rac_head_single(tok, feats[i], threshold=0.5, return_stages=True)
        out[f"{i}:fb:index_down"] = npy(r["index_down"]).astype(np.int32)
        out[f"{i}:fb:idx_cluster"] = npy(r["idx_cluster"]).astype(np.int32)
        out[f"{i}:fb:tokens_rowsum"] = npy(r["tokens"].double().sum(dim=1))
    save("vitl_224", **out)

def gen_tower_features_vitl336():
    """BASELINE cfg4 at full size: reference tower + head for ViT-L/14-**336** (576 patches, T = 577), 2 images, same seeds and recipe as
    vitl_224.  Stores the reference's tower features in full and every downstream output (tokens in full: the dynamic-k threshold yields tens of
    tokens per image)."""
    vc = O.VitConfig(image_size=336)
    hc = O.HeadConfig(threshold=0.125)
    tsd = O.init_tower_weights(vc, seed=0)
    hsd = O.init_head_weights(hc, seed=1)
    d = R.make_clip_dir(vc.hidden_size, vc.num_hidden_layers, vc.num_attention_heads, vc.intermediate_size,
                        vc.image_size, vc.patch_size, seed=0)
    tok = R.build_reference_tokenizer(d, hidden_dim=1024, token_feat_dim=4096, dim_feedforward=4096,
                                      min_cluster_num=64, threshold=0.125, select_layer=-2)
    full = dict(tsd); full.update(hsd)
    load_weights_into(tok, full)
    g = torch.Generator().manual_seed(3)
    images = torch.randn(2, 3, 336, 336, generator=g)
    out = {"spec": np.array([0, 1, 3], dtype=np.int64)}   # tower seed, head seed, image seed
    feats, res = R.rac_forward(tok, images, noise=None, return_stages=True)
    out["feats"] = npy(feats)
    for i, r in enumerate(res):
        print(f"  vitl336/img{i}: L={r['tokens'].shape[0]} score[{r['score'].min():.4f},{r['score'].max():.4f}]")
        out[f"{i}:index_down"] = npy(r["index_down"]).astype(np.int32)
        out[f"{i}:idx_cluster"] = npy(r["idx_cluster"]).astype(np.int32)
        out[f"{i}:score"] = npy(r["score"])
        out[f"{i}:group"] = npy(r["group"]).astype(np.float32)
        out[f"{i}:tokens"] = npy(r["tokens"]).astype(np.float32)
    save("vitl_336", **out)


# ----------------------------------------------------------------------------------------------
DETOK_CASES = {
    # name: (DetokConfig kwargs, seed, token counts per image)
    "small": (dict(token_feat_dim=96, hidden_dim=64, patch_size=14, image_size=70, decoder_embed_dim=64, decoder_nheads=4,
                   decoder_depth=2, num_hidden_layers=4, cross_attention_freq=2, mapper_hidden=64, mapper_heads=4,
                   mapper_intermediate=128), 3, [7, 3, 1, 12]),
    # the Q-Former at bert-base-uncased dims (detokenizer.py:27,80; hidden_dim=768 per train_setokim.py:361), 224^2 / 14 -> 256 queries
    "bertbase": (dict(token_feat_dim=256, hidden_dim=768, patch_size=14, image_size=224, decoder_embed_dim=768, decoder_nheads=16,
                      decoder_depth=1, num_hidden_layers=6, cross_attention_freq=2), 5, [37, 24]),
}


def gen_detok():
    """a9: the reference's Q-Former (BertEmbeddings + BertEncoder, module.py:151-690) on padded tokens + mask, driven as
    BertModel.forward / SetokDeTokenizer.forward drive it.  Weights regenerate bit-exactly from the seed
    (oracle.init_detok_weights); stored: config, seed, the padded inputs, the mask, and the REFERENCE's Q-Former output."""
    arrs = {}
    for name, (kw, seed, counts) in DETOK_CASES.items():
        dc = O.DetokConfig(**kw)
        sd = O.init_detok_weights(dc, seed=seed)
        g = torch.Generator().manual_seed(100 + seed)
        B, L = len(counts), max(counts)
        x = torch.randn(B, L, dc.token_feat_dim, generator=g)
        mask = torch.zeros(B, L)
        for i, c in enumerate(counts):
            mask[i, :c] = 1
        emb, enc = R.build_reference_qformer(hidden=dc.mapper_hidden, heads=dc.mapper_heads, intermediate=dc.mapper_intermediate,
                                             layers=dc.num_hidden_layers, cross_freq=dc.cross_attention_freq,
                                             encoder_width=dc.hidden_dim, num_queries=dc.num_queries, eps=dc.mapper_eps)
        mapped_in = torch.nn.functional.linear(x, sd["mapper_fc_in.weight"], sd["mapper_fc_in.bias"])    # detokenizer.py:104
        ref = R.rac_qformer(emb, enc, sd, sd["mask_tokens"].expand(B, -1, -1), mapped_in, mask)
        arrs[name + ":cfg_keys"] = np.array(list(kw.keys()))
        arrs[name + ":cfg_vals"] = np.array([float(v) for v in kw.values()])
        arrs[name + ":seed"] = np.array(seed)
        arrs[name + ":x"] = npy(x)
        arrs[name + ":mask"] = npy(mask)
        arrs[name + ":mapped_ref"] = npy(ref)
        print(name, "Q-Former reference output", tuple(ref.shape), float(ref.abs().mean()))
    save("detok", **arrs)


# ----------------------------------------------------------------------------------------------
SPLICE_CASES = {
    # name: (seed, B, T, V, D, kwargs)
    "right": (1, 5, 12, 40, 16, dict()),
    "left": (2, 5, 12, 40, 16, dict(padding_side="left")),
    "trunc": (3, 6, 10, 40, 16, dict(max_length=9)),
    "trunc_left": (4, 6, 10, 40, 16, dict(max_length=7, padding_side="left")),
    "nopad_long": (5, 3, 300, 64, 8, dict()),
}


def gen_splice():
    """§8(f) row 1: the reference's own prepare_inputs_labels_for_multimodal (setokim_arch.py:213-355), run unmodified
    (oracle/rac_harness.py::rac_prepare_inputs).  Stored: seeds/specs and the REFERENCE's outputs, with and without the
    optional inputs (position_ids / attention_mask / labels None)."""
    arrs = {}
    for name, (seed, B, T, V, D, kw) in SPLICE_CASES.items():
        ids, am, labels, feats, W = O.splice_inputs(seed, B, T, V, D, pad=not name.startswith("nopad"))
        pos = torch.arange(T).expand(B, T).clone()
        arrs[name + ":spec"] = np.array([seed, B, T, V, D, kw.get("max_length", -1), 1 if kw.get("padding_side") == "left" else 0])
        for variant, (p_, a_, l_) in {"full": (pos, am, labels), "none": (None, None, None)}.items():
            rp, ra, re, rl = R.rac_prepare_inputs(ids, p_, a_, l_, feats, W, **kw)
            arrs[f"{name}:{variant}:embeds"] = npy(re)
            if rp is not None:
                arrs[f"{name}:{variant}:pos"] = npy(rp); arrs[f"{name}:{variant}:mask"] = npy(ra); arrs[f"{name}:{variant}:labels"] = npy(rl)
            else:
                assert ra is None and rl is None
        print(name, "reference output", tuple(re.shape))
    save("splice", **arrs)


# ----------------------------------------------------------------------------------------------
def gen_head_grads():
    """§8(f) row 4: parameter gradients of the reference's head modules by the reference's OWN autograd
    (oracle/rac_harness.py::rac_head_grads) for L = sum_i <tokens_i, upstream_i> on the two dynamic-k images of head_small.npz
    (whose weights and features it reuses).  Stored: the upstream gradients and every parameter gradient."""
    z = np.load(os.path.join(HERE, "head_small.npz"))
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    tok = small_tok()
    res = tok.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys
    feats = [torch.from_numpy(z["dynamic:feats"]), torch.from_numpy(z["planted:feats"])]
    thr = float(z["dynamic:threshold"])
    hc = O.HeadConfig(hidden_dim=64, token_feat_dim=96, min_cluster_num=8, threshold=0.5, nheads=2, dim_feedforward=128)
    with torch.no_grad():
        Ls = [O.head_forward(sd, hc, f, None, thr).tokens.shape[0] for f in feats]
    g = torch.Generator().manual_seed(0)
    ups = [torch.randn(L, 96, generator=g) for L in Ls]
    grads, counts = R.rac_head_grads(tok, feats, ups, threshold=thr)
    assert counts == Ls
    arrs = {"threshold": np.array(thr), "counts": np.array(Ls)}
    for i, u in enumerate(ups):
        arrs[f"up:{i}"] = npy(u)
    for n, v in grads.items():
        arrs["g:" + n] = npy(v)
    print("head grads:", len(grads), "tensors, token counts", Ls)
    save("head_grads", **arrs)


# ----------------------------------------------------------------------------------------------
def gen_bf16_reference(dt=torch.bfloat16, name="bf16_reference"):
    """(dt = torch.float16, name = "fp16_reference": the same yardstick for the fp16 mode, round 6 — the reference's inference loader casts the
    tower to torch.float16, src/model/builder.py:43,135-136.)
    The bf16 throughput mode's yardstick: the REFERENCE's own head cast to torch.bfloat16 (what train_setokim.py:326 does to the whole tower
    module) run on CPU from fixed bf16 features — the reference's fp32 tower features of vitl_224.npz / vitl_336.npz rounded to bf16 — and, per
    stage, its distance from the fp32 oracle evaluated on the SAME bf16-valued inputs (x after the positional add, the bf16-rounded weights) and the
    SAME cluster assignment.  A GPU bf16 result is held to <= 1.5 x these distances (tests/test_fullsize_gpu.py).  Stored per image: the
    reference-bf16 assignment and token count, the three stage errors, the tokens themselves, and what the fp32 clustering yields on the same
    features (the reference's bf16 clustering rounds its scores to bf16 and may pick a different L)."""
    hc = O.HeadConfig(threshold=0.125)
    hsd = O.init_head_weights(hc, seed=1)
    hsd_b = {k: v.to(dt).float() for k, v in hsd.items()}
    d = R.make_clip_dir(1024, 1, 16, 4096, 224, 14, seed=0)            # a 1-layer stand-in tower: only the head is exercised
    tok = R.build_reference_tokenizer(d, hidden_dim=1024, token_feat_dim=4096, dim_feedforward=4096, min_cluster_num=64, threshold=0.125,
                                      select_layer=-2)
    res = tok.load_state_dict(hsd, strict=False)
    assert not res.unexpected_keys
    tok = tok.to(dt)
    low = "bf16" if dt == torch.bfloat16 else "fp16"
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())
    arrs = {}
    for cfg, src in (("cfg2", "vitl_224"), ("cfg4", "vitl_336")):
        feats = torch.from_numpy(np.load(os.path.join(HERE, src + ".npz"))["feats"])
        for i in range(feats.shape[0]):
            r = R.rac_head_single(tok, feats[i].to(dt), return_stages=True)
            x, lab = r["x"].float(), r["idx_cluster"]
            group = O.group_encoding(hsd_b, hc, x, lab)
            inter = O.block_forward(hsd_b, "inter_encoder.", group, hc.nheads, hc.intra_cluster_layers)
            tokens = torch.nn.functional.linear(inter, hsd_b["out.weight"], hsd_b["out.bias"])
            errs = [rel(r["group"].float(), group), rel(r["inter"].float(), inter), rel(r["tokens"].float(), tokens)]
            f32 = O.cluster_dpc_knn(x, hc.min_cluster_num, hc.threshold, hc.min_cluster_num)
            pre = f"{cfg}:{i}:"
            arrs[pre + "idx_cluster"] = npy(lab).astype(np.int32)
            arrs[pre + "errs"] = np.array(errs)
            arrs[pre + "L"] = np.array([r["tokens"].shape[0], f32.index_down.numel()])
            if cfg == "cfg2":
                arrs[pre + "tokens"] = npy(r["tokens"].float())
            print(f"  {cfg}/img{i}: reference-{low} L = {r['tokens'].shape[0]} (fp32 clustering of the same features: {f32.index_down.numel()}); "
                  f"reference-{low} vs fp32 oracle: group {errs[0]:.3e} inter {errs[1]:.3e} tokens {errs[2]:.3e}")
    save(name, **arrs)


def partition_drift(index_down_a, idx_a, index_down_b, idx_b):
    """How far two clusterings of the same image are apart: (|L_a - L_b|, 1 - Jaccard of the centre-token sets, fraction of tokens whose
    centre TOKEN differs — label ids shift with the centre list, centre tokens do not)."""
    a, b = set(index_down_a.tolist()), set(index_down_b.tolist())
    jacc = len(a & b) / float(len(a | b))
    same = float((index_down_a[idx_a] == index_down_b[idx_b]).float().mean())
    return abs(len(a) - len(b)), 1.0 - jacc, 1.0 - same


def gen_bf16_tower(dt=torch.bfloat16, name="bf16_tower"):
    """(dt = torch.float16, name = "fp16_tower": the fp16 mode's yardstick, round 6; its keys say `lowbits` where the bf16 fixture says `bf16_bits`,
    and the ViT-L feature bits are not stored — errors and drifts only.)
    VERDICT r03 item 5 — the throughput mode's yardstick FROM PIXELS: the reference's own tower (HF CLIPVisionModel behind the reference's
    CLIPVisionTower) AND head, cast to torch.bfloat16 as train_setokim.py:326 casts the module, run on CPU on the images of vitl_224.npz
    (ViT-L/14-224, seeds 0 / 1 / 3) and of e2e_small.npz's recipe; stored: the reference-bf16 tower features (bf16 bit patterns), their
    distance from the reference's fp32 features, and how far the reference's bf16 clustering drifts from its fp32 clustering (token count,
    centre set, partition).  tests/test_fullsize_gpu.py holds the GPU's bf16 tower and clustering to 1.5 x these drifts."""
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())
    rms = lambda a, b: float(((a.double() - b.double()) ** 2).mean().sqrt() / (b.double() ** 2).mean().sqrt())
    arrs = {}
    is_bf = dt == torch.bfloat16
    low, bits_key = ("bf16", "feats_bf16_bits") if is_bf else ("fp16", "feats_lowbits")
    # ---- ViT-L/14-224, 2 images (the inputs and fp32 outputs of vitl_224.npz) ----------------------------------------------------------
    vc, hc = O.VitConfig(), O.HeadConfig(threshold=0.125)
    tsd, hsd = O.init_tower_weights(vc, seed=0), O.init_head_weights(hc, seed=1)
    d = R.make_clip_dir(vc.hidden_size, vc.num_hidden_layers, vc.num_attention_heads, vc.intermediate_size, vc.image_size, vc.patch_size, seed=0)
    z = np.load(os.path.join(HERE, "vitl_224.npz"))
    feats32 = torch.from_numpy(z["feats"])
    images = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(3))
    for sel in (-2, -1):
        tok = R.build_reference_tokenizer(d, hidden_dim=1024, token_feat_dim=4096, dim_feedforward=4096, min_cluster_num=64, threshold=0.125,
                                          select_layer=sel)
        full = dict(tsd); full.update(hsd)
        load_weights_into(tok, full)
        if sel == -2:
            f32_feats, f32_res = feats32, [dict(index_down=torch.from_numpy(z[f"{i}:index_down"]).long(),
                                                idx_cluster=torch.from_numpy(z[f"{i}:idx_cluster"]).long()) for i in range(2)]
        else:                                                        # the launch scripts' layer selection: fp32 run here, features stored too
            f32_feats, f32_res = R.rac_forward(tok, images, noise=None, return_stages=True)
            if is_bf:                                                # (the fp16 fixture reads the fp32 run of select_layer = -1 from bf16_tower.npz)
                arrs["vitl:sel-1:feats32"] = npy(f32_feats)
                for i, r in enumerate(f32_res):
                    arrs[f"vitl:sel-1:{i}:index_down32"] = npy(r["index_down"]).astype(np.int32)
                    arrs[f"vitl:sel-1:{i}:idx_cluster32"] = npy(r["idx_cluster"]).astype(np.int32)
        tok_b = tok.to(dt)
        feats_b, res_b = R.rac_forward(tok_b, images.to(dt), noise=None, return_stages=True)
        tag = f"vitl:sel{sel}"
        if is_bf:
            arrs[tag + ":feats_bf16_bits"] = npy(feats_b.view(torch.int16))
        arrs[tag + ":tower_err"] = np.array([rel(feats_b.float(), f32_feats), rms(feats_b.float(), f32_feats)])
        print(f"  {tag}: reference-{low} tower vs its fp32 tower: max-rel {rel(feats_b.float(), f32_feats):.3e}  rms-rel {rms(feats_b.float(), f32_feats):.3e}")
        for i, r in enumerate(res_b):
            dl, dj, dp = partition_drift(r["index_down"], r["idx_cluster"], f32_res[i]["index_down"], f32_res[i]["idx_cluster"])
            arrs[f"{tag}:{i}:drift"] = np.array([dl, dj, dp])
            arrs[f"{tag}:{i}:L"] = np.array([r["index_down"].numel(), f32_res[i]["index_down"].numel()])
            print(f"  {tag}/img{i}: reference-{low} L = {r['index_down'].numel()} (its fp32 run: {f32_res[i]['index_down'].numel()}); "
                  f"1 - centre Jaccard {dj:.3f}; tokens with another centre {dp:.3f}")
    # ---- small dims, the whole path (the tower of e2e_small's recipe), 4 images --------------------------------------------------------
    tok = small_tok(sel=-2)
    sdz = np.load(os.path.join(HERE, "e2e_small.npz"))
    sd = {k[2:]: torch.from_numpy(sdz[k]) for k in sdz.files if k.startswith("w:")}
    if sd:
        load_weights_into(tok, sd)
    images = torch.randn(4, 3, 112, 112, generator=torch.Generator().manual_seed(21))
    thr = 0.5
    f32_feats, f32_res = R.rac_forward(tok, images, threshold=thr, noise=None, return_stages=True)
    arrs["small:w_keys"] = np.array(sorted(tok.state_dict().keys()))
    for k_, v_ in tok.state_dict().items():
        arrs["small:w:" + k_] = npy(v_)
    tok_b = tok.to(dt)
    feats_b, res_b = R.rac_forward(tok_b, images.to(dt), threshold=thr, noise=None, return_stages=True)
    arrs["small:feats32"] = npy(f32_feats)
    arrs["small:" + bits_key] = npy(feats_b.view(torch.int16))
    arrs["small:tower_err"] = np.array([rel(feats_b.float(), f32_feats), rms(feats_b.float(), f32_feats)])
    if not is_bf:                                                    # the whole path at small dims in fp16: the reference's tokens and its clustering drift from fp32
        for i, (rb, r32) in enumerate(zip(res_b, f32_res)):
            dl, dj, dp = partition_drift(rb["index_down"], rb["idx_cluster"], r32["index_down"], r32["idx_cluster"])
            arrs[f"small:{i}:drift"] = np.array([dl, dj, dp])
            arrs[f"small:{i}:L"] = np.array([rb["index_down"].numel(), r32["index_down"].numel()])
            arrs[f"small:{i}:tokens32"] = npy(r32["tokens"])
            arrs[f"small:{i}:idx_cluster32"] = npy(r32["idx_cluster"]).astype(np.int32)
            if rb["tokens"].shape == r32["tokens"].shape:
                arrs[f"small:{i}:tokens_err"] = np.array([rel(rb["tokens"].float(), r32["tokens"]), rms(rb["tokens"].float(), r32["tokens"])])
            print(f"  small/img{i}: reference-{low} L = {rb['index_down'].numel()} (fp32 {r32['index_down'].numel()}), drift {dl} {dj:.3f} {dp:.3f}")
    print(f"  small: reference-{low} tower vs fp32: max-rel {rel(feats_b.float(), f32_feats):.3e}  rms-rel {rms(feats_b.float(), f32_feats):.3e}")
    save(name, **arrs)


STAGE2_CASES = {
    # name: (projector_type, seed, B, T, V, token_feat_dim, hidden, kwargs, train_embed)
    "mlp2x": ("mlp2x_gelu", 11, 5, 12, 40, 96, 64, dict(), False),
    "linear_trunc": ("linear", 12, 6, 10, 40, 96, 64, dict(max_length=9), False),
    "mlp2x_norm_left_embed": ("mlp2x_gelu_Norm", 13, 4, 14, 48, 96, 64, dict(padding_side="left"), True),
}


def gen_stage2():
    """Stage 2 of the reference's recipe (scripts/pretrain_mm_proj.sh:40): the REFERENCE's build_vision_projector module + the REFERENCE's
    prepare_inputs_labels_for_multimodal under torch autograd (oracle/rac_harness.py::rac_stage2_grads), a small downstream loss standing in
    for the LLM.  Stored: the projector's initial weights, the tokens, the downstream matrix, and the reference's loss / inputs_embeds /
    projector gradients / d loss / d tokens (/ d embed_tokens.weight)."""
    B_ = R.load_reference_projector_builder()
    arrs = {}
    for name, (ptype, seed, B, T, V, Dt, Dh, kw, train_embed) in STAGE2_CASES.items():
        ids, am, labels, feats, W = O.splice_inputs(seed, B, T, V, Dh)
        g = torch.Generator().manual_seed(seed + 1000)
        toks = [torch.randn(f.shape[0], Dt, generator=g) for f in feats]
        w_down = torch.randn(V, Dh, generator=g) * 0.3
        torch.manual_seed(seed)
        proj = B_.build_vision_projector(ptype, mm_hidden_size=Dt, hidden_size=Dh)
        with torch.no_grad():
            for p_ in proj.parameters():                       # non-trivial biases / LayerNorm affines
                if p_.dim() == 1:
                    p_.add_(0.1 * torch.randn(p_.shape, generator=g))
        pos = torch.arange(T).expand(B, T).clone()
        w0 = {n: p_.detach().clone() for n, p_ in proj.named_parameters()}
        loss, embeds, new_labels, pg, tg, eg = R.rac_stage2_grads(proj, toks, ids, pos, am, labels, W, w_down, train_embed=train_embed, **kw)
        arrs[name + ":spec"] = np.array([seed, B, T, V, Dt, Dh, kw.get("max_length", -1), 1 if kw.get("padding_side") == "left" else 0, int(train_embed)])
        arrs[name + ":ptype"] = np.array(ptype)
        arrs[name + ":w_down"] = npy(w_down)
        arrs[name + ":loss"] = npy(loss)
        arrs[name + ":embeds"] = npy(embeds)
        arrs[name + ":counts"] = np.array([t.shape[0] for t in toks])
        arrs[name + ":tokens"] = npy(torch.cat(toks, 0))
        arrs[name + ":dtokens"] = npy(torch.cat(tg, 0))
        for n, v in w0.items():
            arrs[f"{name}:w:{n}"] = npy(v)
        for n, v in pg.items():
            arrs[f"{name}:g:{n}"] = npy(v)
        if eg is not None:
            arrs[name + ":dembed"] = npy(eg)
        print(name, "loss", float(loss), "embeds", tuple(embeds.shape), "grads", sorted(pg), "zero token-grad rows",
              int((torch.cat(tg, 0).abs().sum(1) == 0).sum()))
    save("stage2", **arrs)


LLAMA_CASES = {
    # name: (LlamaConfigLite kwargs, seed, B, T, padding)
    "tiny_right": (dict(hidden_size=64, intermediate_size=176, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4, vocab_size=100), 7, 3, 11, "right"),
    "tiny_left": (dict(hidden_size=64, intermediate_size=176, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4, vocab_size=100), 8, 3, 11, "left"),
    "dh128": (dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2, vocab_size=128), 9, 2, 150, "right"),
    "dh128_left": (dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2, vocab_size=128), 10, 2, 70, "left"),
    # grouped-query attention (num_key_value_heads < num_attention_heads: HF repeat_kv)
    "gqa_tiny_left": (dict(hidden_size=64, intermediate_size=176, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, vocab_size=100), 11, 3, 13, "left"),
    "gqa_dh128": (dict(hidden_size=512, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, vocab_size=128), 12, 2, 150, "right"),
    "mqa_dh128_left": (dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1, vocab_size=128), 13, 2, 70, "left"),
}


def gen_llama():
    """cfg 5 (SURVEY.md 8f last row): HuggingFace LlamaForCausalLM (the `self.model` / `self.lm_head` of setokim_llama.py:130-143; third
    party, installed transformers, eager attention, fp32) on seeded inputs_embeds / attention_mask / position_ids.  Weights regenerate from
    the seed (oracle.init_llama_weights); stored: the HF logits and final hidden states."""
    from transformers import LlamaConfig, LlamaForCausalLM
    arrs = {}
    for name, (kw, seed, B, T, padding) in LLAMA_CASES.items():
        lc = O.LlamaConfigLite(**kw)
        sd = O.init_llama_weights(lc, seed=seed)
        cfg = LlamaConfig(**kw, rms_norm_eps=lc.rms_norm_eps, rope_theta=lc.rope_theta, attention_bias=False, mlp_bias=False, tie_word_embeddings=False)
        cfg._attn_implementation = "eager"
        m = LlamaForCausalLM(cfg).eval()
        m.load_state_dict(sd, strict=True)
        x, am, pos = O.llama_inputs(lc, seed, B, T, padding)
        with torch.no_grad():
            out = m(inputs_embeds=x, attention_mask=am, position_ids=pos, output_hidden_states=True)
        arrs[name + ":cfg_keys"] = np.array(list(kw.keys())); arrs[name + ":cfg_vals"] = np.array(list(kw.values()))
        arrs[name + ":spec"] = np.array([seed, B, T, 1 if padding == "left" else 0])
        arrs[name + ":logits"] = npy(out.logits); arrs[name + ":hidden"] = npy(out.hidden_states[-1])
        print(name, "HF logits", tuple(out.logits.shape))
    save("llama", **arrs)


LLAMA_7B_DIMS = dict(hidden_size=4096, intermediate_size=11008, num_hidden_layers=2, num_attention_heads=32, num_key_value_heads=32, vocab_size=32000)
LLAMA_7B_CASE = (21, 2, 40, "right")               # seed, B, T, padding
LOGIT_COL_STRIDE = 61                               # every 61st vocabulary column at every position (+ all 32000 at each sequence's last token)


def gen_llama_7bdims():
    """BASELINE cfg5 at Vicuna-7B layer dims (hidden 4096, 32 heads x 128, SwiGLU 11008, vocab 32000), TWO decoder layers: HuggingFace
    LlamaForCausalLM (eager attention, fp32) on seeded inputs.  The 0.67 G parameters regenerate from the seed (oracle.init_llama_weights);
    stored: the final hidden states in full, the logits at every position for every 61st vocabulary column, and all 32000 logits at each
    sequence's last valid token."""
    from transformers import LlamaConfig, LlamaForCausalLM
    kw = LLAMA_7B_DIMS
    seed, B, T, padding = LLAMA_7B_CASE
    lc = O.LlamaConfigLite(**kw)
    sd = O.init_llama_weights(lc, seed=seed)
    cfg = LlamaConfig(**kw, rms_norm_eps=lc.rms_norm_eps, rope_theta=lc.rope_theta, attention_bias=False, mlp_bias=False, tie_word_embeddings=False)
    cfg._attn_implementation = "eager"
    with torch.device("meta"):
        m = LlamaForCausalLM(cfg)
    m = m.to_empty(device="cpu").eval()
    m.load_state_dict(sd, strict=True, assign=True)
    # rotary inv_freq is a non-persistent buffer: to_empty() left it uninitialised -> rebuild it the way HF does
    inv = 1.0 / (lc.rope_theta ** (torch.arange(0, lc.head_dim, 2, dtype=torch.int64).float() / lc.head_dim))
    m.model.rotary_emb.inv_freq = inv
    if hasattr(m.model.rotary_emb, "original_inv_freq"):
        m.model.rotary_emb.original_inv_freq = inv
    x, am, pos = O.llama_inputs(lc, seed, B, T, padding)
    with torch.no_grad():
        out = m(inputs_embeds=x, attention_mask=am, position_ids=pos, output_hidden_states=True)
        ref_h, ref_l = O.llama_forward(sd, lc, x, am, pos)
    print("oracle vs HF at 7B dims: hidden", float((ref_h - out.hidden_states[-1]).abs().max()), "logits", float((ref_l - out.logits).abs().max()))
    last = [int(am[b].nonzero().max()) for b in range(B)]
    arrs = {"cfg_keys": np.array(list(kw.keys())), "cfg_vals": np.array(list(kw.values())), "spec": np.array([seed, B, T, 0]),
            "hidden": npy(out.hidden_states[-1]), "logits_cols": npy(out.logits[:, :, ::LOGIT_COL_STRIDE]),
            "logits_last": npy(torch.stack([out.logits[b, last[b]] for b in range(B)])), "last": np.array(last)}
    save("llama_7bdims", **arrs)


LM_LOSS_CASES = {"plain": (0, 2, 9, 37, "none"), "right_pad": (1, 3, 12, 50, "right"), "left_pad_ignored": (2, 3, 11, 64, "left")}


def gen_lm_loss():
    """The language-model loss: the reference's own statements (setokim_llama.py:145-160) executed by rac_harness.rac_lm_loss on seeded
    logits / labels / masks.  Inputs regenerate from the seed (oracle.lm_loss_inputs); stored: the loss."""
    arrs = {}
    for name, (seed, B, T, V, padding) in LM_LOSS_CASES.items():
        logits, labels, am = O.lm_loss_inputs(seed, B, T, V, padding)
        loss = R.rac_lm_loss(logits, labels, am)
        arrs[name + ":spec"] = np.array([seed, B, T, V]); arrs[name + ":padding"] = np.array(padding)
        arrs[name + ":loss"] = npy(loss.reshape(1))
        print(name, "loss", float(loss))
    save("lm_loss", **arrs)


if __name__ == "__main__":
    which = sys.argv[1:] or ["head_small", "cluster_full", "e2e_small", "vitl", "detok", "splice", "head_grads", "llama", "lm_loss", "stage2"]
    if "stage2" in which:
        gen_stage2()
    if "bf16_reference" in which:
        gen_bf16_reference()
    if "bf16_tower" in which:
        gen_bf16_tower()
    if "fp16_reference" in which:
        gen_bf16_reference(torch.float16, "fp16_reference")
    if "fp16_tower" in which:
        gen_bf16_tower(torch.float16, "fp16_tower")
    if "detok" in which:
        gen_detok()
    if "splice" in which:
        gen_splice()
    if "head_grads" in which:
        gen_head_grads()
    if "llama" in which:
        gen_llama()
    if "lm_loss" in which:
        gen_lm_loss()
    if "head_small" in which:
        gen_head_small()
    if "cluster_full" in which:
        gen_cluster_full()
    if "e2e_small" in which:
        gen_e2e_small()
    if "vitl" in which:
        gen_tower_features_vitl()
    if "vitl336" in which:
        gen_tower_features_vitl336()
    if "llama_7bdims" in which:
        gen_llama_7bdims()
