"""float16 as the third element type (round 6; VERDICT r05 "missing" 3): the reference's inference loader moves the tower to torch.float16
(/root/reference/src/model/builder.py:43,135-136) and so does every non-`--bf16` training launch (src/train/train_setokim.py:326,348,374) — a
caller that does what the reference's own loader does must get features, not a TypeError.

The 16-bit kernels are compiled twice from one source (setok_amd/csrc/common.h, SETOK_HALF): libsetok_hip_f16.so is the fp16 build, the Python
host routes every call whose tensors are float16 to it (setok_amd/_lib.py).  Here: every kernel class of the fp16 build against fp32 torch on the
same fp16-valued operands, the whole path against the op-by-op host path and the oracle, the reference-style loader sequence, and the properties
the bf16 mode has (determinism, batch invariance).  The yardstick tests against the reference's OWN fp16 run (tests/golden/fp16_reference.npz,
fp16_tower.npz) are the `low = "fp16"` cases of tests/test_fullsize_gpu.py.  Needs a real MI355X: `pytest -m gpu`."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import setok_oracle as O

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from setok_amd import SetokTokenizer, _lib, ops

DEV = "cuda"
H = torch.float16


def _rand(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _rel(got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def test_fp16_calls_run_in_the_fp16_build_and_nowhere_else():
    """The routing: a float16 call loads libsetok_hip_f16.so; a bfloat16 / float32 call never touches it; the bf16 build refuses fp16 buffers."""
    a, w = _rand(64, 64, seed=1).to(DEV, H), _rand(64, 64, seed=2).to(DEV, H)
    out = ops.linear(a, w)
    assert out.dtype == H and True in _lib._libs
    with pytest.raises(_lib.SetokHipError):
        _lib.call("setok_layernorm", ops._stream(), 2, a.data_ptr(), w.data_ptr(), w.data_ptr(), out.data_ptr(), 64, 64, 1e-5)   # the fp16 code (a plain int: no routing) at the bf16 build
    with pytest.raises(TypeError):
        ops._code(torch.float64)


# ---------------------------------------------------------------------------------------------
# GEMM classes: small-tile kernels, the persistent 256 x 256 ping-pong kernel (plain / residual / LayerNorm folded), fp32 output
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K,act,res", [(128, 128, 64, 0, False), (257, 192, 128, 1, True), (1, 96, 64, 2, False), (514, 3072, 1024, 1, True), (77, 64, 640, 0, True),
                                           (8192, 4096, 1024, 1, False), (7776, 3072, 768, 2, False), (7680, 768, 3072, 0, True), (6000, 1024, 4096, 0, True),
                                           (4100, 2112, 192, 2, True), (25700, 1344, 320, 0, False), (65792, 1024, 128, 0, True)])
def test_linear_fp16(M, N, K, act, res):
    a, w = _rand(M, K, seed=1).to(H), _rand(N, K, seed=2, scale=K ** -0.5).to(H)
    b = _rand(N, seed=3)
    r = _rand(M, N, seed=4).to(H) if res else None
    ref = F.linear(a.double(), w.double(), b.double())
    ref = [ref, O.quick_gelu(ref), F.gelu(ref)][act]
    if res:
        ref = ref.to(H).double() + r.double()              # torch's 16-bit semantics: the Linear output is rounded, then the residual is added
    got = ops.linear(a.to(DEV), w.to(DEV), b.to(DEV), None if r is None else r.to(DEV), act=act)
    assert got.dtype == H
    err = (got.double().cpu() - ref).abs()
    assert _rel(got, ref) < 1e-3                            # one fp16 rounding of the result (2^-11) + fp32 accumulation
    assert bool((err <= 2.0 ** -10 * ref.abs() + 4e-3).all())            # element-wise: one fp16 ulp + accumulation-order noise (erf-GELU: the A&S polynomial, 1.5e-7)
    if not res:
        got32 = ops.linear(a.to(DEV), w.to(DEV), b.to(DEV), act=act, out_dtype=torch.float32)
        assert _rel(got32, ref) < 2e-5                      # fp16 products are exact in fp32; only the accumulation order differs


@pytest.mark.parametrize("N,K", [(1024, 1024), (4096, 1024), (1024, 4096), (2368, 448)])
@pytest.mark.parametrize("act,use_res", [(0, False), (1, False), (2, True), (0, True)])
def test_linear_fp16_rows_do_not_depend_on_the_kernel(act, use_res, N, K):
    """Batch invariance holds in the fp16 build as in the bf16 one: the same rows through the persistent kernel and every small-tile shape give the same bits.
    (N = 2368 is not a multiple of 256: the round-2 persistent kernel.  Its accumulators started at the bias through the matrix pipe from an "exact" three-way
    16-bit split — exact in bf16, which has fp32's exponent range, not in fp16, where the third part of a bias of order 1 is a subnormal with two bits left: 8e-5 of
    its outputs were one fp16 ulp off the other kernels'.  Found by tools/fuzz_gpu.py ... f16, seed 7; the fp16 build starts at the fp32 bias directly.)"""
    M = 25088
    a, w = _rand(M, K, seed=11).to(DEV, H), _rand(N, K, seed=12, scale=K ** -0.5).to(DEV, H)
    b = _rand(N, seed=13).to(DEV)
    r = _rand(M, N, seed=14).to(DEV, H) if use_res else None
    big = ops.linear(a, w, b, r, act=act)
    for m in (1, 63, 257, 1028, 2056, 4112):
        small = ops.linear(a[:m].contiguous(), w, b, None if r is None else r[:m].contiguous(), act=act)
        assert torch.equal(small, big[:m]), m


@pytest.mark.parametrize("M,N,K,act", [(65792 // 8, 3072, 1024, 0), (8224, 4096, 1024, 1), (300, 1024, 1024, 0), (257, 3072, 768, 1), (2056, 2304, 768, 2)])
def test_layernorm_folded_linear_fp16(M, N, K, act):
    """setok_row_stats + setok_ln_fold + setok_linear_ln in the fp16 build (the two-way splits by rounding, common.h split2): against
    LayerNorm -> Linear in fp64 on the same fp16-valued x, W, and the same rows whether the persistent or the small-tile kernel produced them."""
    x = (_rand(M, K, seed=1) * 1.7 + 0.3).to(H)
    w, b = _rand(N, K, seed=2, scale=K ** -0.5).to(H), _rand(N, seed=3)
    gamma, beta = 1.0 + 0.1 * _rand(K, seed=4), 0.1 * _rand(K, seed=5)
    ref = F.linear(F.layer_norm(x.double(), (K,), gamma.double(), beta.double(), 1e-5), w.double(), b.double())
    ref = [ref, O.quick_gelu(ref), F.gelu(ref)][act]
    folded = ops.ln_fold(w.to(DEV), gamma.to(DEV), beta.to(DEV), b.to(DEV))
    assert folded[0].dtype == H
    xd = x.to(DEV)
    got = ops.linear_ln(xd, folded, ops.row_stats(xd, 1e-5), act=act)
    assert got.dtype == H
    # W' = fp16(gamma W) is rounded once more than the unfolded form: 2^-11 on every product, averaging out over K
    assert _rel(got, ref) < 2.5e-3, _rel(got, ref)
    m = min(M, 257)
    part = ops.linear_ln(xd[:m].contiguous(), folded, ops.row_stats(xd[:m].contiguous(), 1e-5), act=act)
    assert torch.equal(part, got[:m])


# ---------------------------------------------------------------------------------------------
# attention, LayerNorm, clustering, glue
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,T,Hh,Dh", [(3, 257, 16, 64), (2, 577, 16, 64), (2, 197, 12, 64), (2, 324, 16, 48), (1, 50, 4, 64)])
def test_vit_attention_fp16(B, T, Hh, Dh):
    C = Hh * Dh
    qkv = _rand(B * T, 3 * C, seed=1).to(H)
    q, k, v = [t.reshape(B, T, Hh, Dh).transpose(1, 2).double() for t in qkv.split(C, dim=1)]
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B * T, C)
    got = ops.attention(qkv.to(DEV), Hh, Dh, Dh ** -0.5, T)
    assert got.dtype == H and _rel(got, ref) < 2e-3


def test_segment_attention_layernorm_and_glue_fp16():
    """The head's ragged attention (2 heads x 512), LayerNorm rows, gather / segment mean in the fp16 build against fp64 torch."""
    C, Hh = 1024, 2
    lens = [1, 3, 40, 7, 33, 64, 2, 9, 100]
    offs = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32)
    R = int(offs[-1])
    qkv = _rand(R, 3 * C, seed=2, scale=0.5).to(H)
    got = ops.attention(qkv.to(DEV), Hh, C // Hh, (C // Hh) ** -0.5, max(lens), seg_offsets=offs.to(DEV), n_segs=len(lens)).float().cpu()
    for i, n in enumerate(lens):
        s = slice(int(offs[i]), int(offs[i + 1]))
        q, k, v = [t.reshape(n, Hh, C // Hh).transpose(0, 1).double() for t in qkv[s].split(C, dim=1)]
        ref = F.scaled_dot_product_attention(q, k, v).transpose(0, 1).reshape(n, C)
        assert _rel(got[s], ref) < 2e-3, (i, n)
    x = (_rand(700, C, seed=3) * 2 + 0.5).to(H)
    g, b = 1 + 0.1 * _rand(C, seed=4), 0.1 * _rand(C, seed=5)
    y = ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), 1e-5)
    assert y.dtype == H and _rel(y, F.layer_norm(x.double(), (C,), g.double(), b.double(), 1e-5)) < 1e-3
    perm = torch.randperm(700, generator=torch.Generator().manual_seed(6)).int()
    assert torch.equal(ops.gather_rows(x.to(DEV), perm.to(DEV)).cpu(), x[perm.long()])


@pytest.mark.parametrize("N,grid", [(256, 16), (576, 24), (196, 14)])
def test_clustering_fp16_decisions_equal_the_oracles_where_certain(N, grid):
    """cluster_dpc_knn in the fp16 build (one launch at N <= 256, strips at N <= 576; the Gram product on v_mfma_f32_*_f16): d^2 from exact fp16
    products with fp32 accumulation — the reference's formula up to summation order — so every decision the fp64 margin analysis calls certain
    equals the fp32 oracle's on the same fp16-valued features (tokenizer.py:78-121)."""
    C, B = 1024, 3
    x = torch.stack([O.planted_features(N, C, 8 + 4 * i, seed=20 + i) for i in range(B)]).to(H)
    k, thr, mc = 8, 0.5, 8
    idx, score, index_down, counts = ops.cluster_dpc_knn(x.reshape(B * N, C).to(DEV), B, N, k, thr, mc)
    for i in range(B):
        ref = O.cluster_dpc_knn(x[i].float(), k, thr, mc)
        sens = O.cluster_sensitivity(x[i].float(), k, thr, mc, ulps=16.0)
        L = int(counts[i])
        O.check_cluster_parity(index_down[i, :L].cpu(), idx[i].cpu(), ref.index_down, ref.idx_cluster, sens)        # raises on a certain mismatch


# ---------------------------------------------------------------------------------------------
# the whole path
# ---------------------------------------------------------------------------------------------
def _small_tok(dtype, sd=None, sel=-2):
    vc = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=4, image_size=112, patch_size=14)
    tok = SetokTokenizer(vision_tower=vc, mm_vision_select_layer=sel, hidden_dim=64, token_feat_dim=96, min_cluster_num=8, threshold=0.5, nheads=2, dim_feedforward=128)
    if sd is not None:
        assert not tok.load_state_dict(sd, strict=False).unexpected_keys
    return tok.to(device=DEV, dtype=dtype).eval()


def test_the_reference_loaders_sequence_in_fp16(golden_dir):
    """What src/model/builder.py:134-138 does with the tower: build it, load_model(), `.to(device=..., dtype=torch.float16)`, then call it on
    float16 images (the image processor's output cast by the caller, src/model/setok/clip_encoder.py:55,59).  Round 5 raised TypeError at the
    first kernel.  The features come back in float16, the one setok_encode call equals the op-by-op host path bit for bit, an image's result
    does not depend on its batch, and the tokens sit within 1.5 x the reference's OWN fp16 run of the fp32 reference (tests/golden/fp16_tower.npz)."""
    zt = np.load(os.path.join(golden_dir, "fp16_tower.npz"))
    sd = {k[len("small:w:"):]: torch.from_numpy(zt[k]) for k in zt.files if k.startswith("small:w:")}
    tok = _small_tok(torch.float32, sd)
    if hasattr(tok, "load_model"):
        tok.load_model()
    tok = tok.to(device=DEV, dtype=torch.float16)                                    # builder.py:135-136
    assert tok.dtype == torch.float16
    images = torch.randn(4, 3, 112, 112, generator=torch.Generator().manual_seed(21))
    with torch.no_grad():
        feats, idx, score = tok(images.to(DEV, torch.float16), threshold=0.5)
    assert feats.packed.dtype == torch.float16 and score[0].dtype == torch.float32 and idx[0].dtype == torch.int64
    os.environ["SETOK_HOST_PATH"] = "1"
    try:
        with torch.no_grad():
            f2, i2, s2 = tok(images.to(DEV, torch.float16), threshold=0.5)
    finally:
        del os.environ["SETOK_HOST_PATH"]
    assert feats.counts == f2.counts and torch.equal(feats.packed, f2.packed) and torch.equal(idx, i2) and torch.equal(score, s2)
    with torch.no_grad():
        f1, i1, s1 = tok(images[2:3].to(DEV, torch.float16), threshold=0.5)
    assert torch.equal(f1[0], feats[2]) and torch.equal(i1[0], idx[2])              # alone = inside the batch
    rms = lambda a, b: float(((a.double() - b.double()) ** 2).mean().sqrt() / (b.double() ** 2).mean().sqrt())
    checked = 0
    for i in range(4):
        want, lab32 = torch.from_numpy(zt[f"small:{i}:tokens32"]), torch.from_numpy(zt[f"small:{i}:idx_cluster32"]).long()
        key = f"small:{i}:tokens_err"
        if key not in zt.files or not torch.equal(idx[i].cpu(), lab32):
            continue                                                                   # the reference's own fp16 run (or ours) clusters this image differently: no token-wise statement
        ref_max, ref_rms = zt[key].tolist()
        got_max, got_rms = _rel(feats[i].float(), want), rms(feats[i].float().cpu(), want)
        print(f"small dims fp16 tokens, image {i}: GPU vs reference-fp32 max-rel {got_max:.3e} rms-rel {got_rms:.3e}; reference-fp16 {ref_max:.3e} / {ref_rms:.3e}")
        assert got_max <= 1.5 * ref_max and got_rms <= 1.5 * ref_rms
        checked += 1
    assert checked >= 2


def test_fp16_encode_is_deterministic_and_graph_replay_equals_eager():
    from setok_amd.context import GraphedEncode
    vc = O.VitConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=4, image_size=112, patch_size=14)
    hc = O.HeadConfig(hidden_dim=64, token_feat_dim=128, min_cluster_num=8, threshold=0.5, nheads=2, dim_feedforward=128)
    sd = O.init_tower_weights(vc, seed=0)
    sd.update(O.init_head_weights(hc, seed=1))
    tok = SetokTokenizer(vision_tower=vars(vc), hidden_dim=64, token_feat_dim=128, min_cluster_num=8, threshold=0.5, nheads=2, dim_feedforward=128)
    tok.load_state_dict(sd, strict=False)
    tok = tok.to(DEV, H).eval()
    ctx = tok._context()
    g = GraphedEncode(ctx, 3, threshold=0.16)
    for seed in (0, 1):
        images = torch.randn(3, 3, 112, 112, generator=torch.Generator().manual_seed(seed)).to(DEV, H)
        a, a2, b = ctx.encode(images, threshold=0.16), ctx.encode(images, threshold=0.16), g(images)
        assert a[1] == a2[1] == b[1] and torch.equal(a[0], a2[0]) and torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    # fp16 sits closer to the fp32 oracle than bf16 does (11 against 8 significand bits) — a sanity check that the build really computes in fp16
    images = torch.randn(3, 3, 112, 112, generator=torch.Generator().manual_seed(0))
    feats32 = O.tower_forward(O.normalise_tower_keys(sd), vc, images, hc.mm_vision_select_layer, hc.mm_vision_select_feature)
    f16 = tok.image_feature_encoder(images.to(DEV, H)).float().cpu()
    fb = tok.to(DEV, torch.bfloat16).image_feature_encoder(images.to(DEV, torch.bfloat16)).float().cpu()
    e16, eb = _rel(f16, feats32), _rel(fb, feats32)
    print(f"tower features vs fp32 oracle: fp16 {e16:.3e}, bf16 {eb:.3e}")
    assert e16 < 0.5 * eb


def test_projector_in_fp16():
    """encode_images' second half (setokim_arch.py:206-211, multimodal_projector/builder.py:33-59) on float16 tokens: mlp2x_gelu vs fp64 torch."""
    from setok_amd.builder import build_vision_projector
    proj = build_vision_projector("mlp2x_gelu", mm_hidden_size=128, hidden_size=192).to(DEV, H).eval()
    x = _rand(37, 128, seed=3).to(H)
    with torch.no_grad():
        y = proj(x.to(DEV))
    lin = [m for m in proj.modules() if isinstance(m, torch.nn.Linear)]
    ref = F.linear(F.gelu(F.linear(x.double(), lin[0].weight.double().cpu(), lin[0].bias.double().cpu())).to(H).double(), lin[1].weight.double().cpu(), lin[1].bias.double().cpu())
    assert y.dtype == H and _rel(y, ref) < 2e-3
