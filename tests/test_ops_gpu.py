"""Per-operator parity of the HIP library (through the C ABI) against the CPU oracle / plain torch
fp32 restatements of the same op.  Needs a real MI355X: `pytest -m gpu`."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import setok_oracle as O

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from setok_amd import ops

DEV = "cuda"


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _rel_err(got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


# ---------------------------------------------------------------------------------------------
# setok_linear
# ---------------------------------------------------------------------------------------------
LIN_SHAPES = [(128, 128, 64), (257, 192, 128), (1, 96, 64), (300, 1024, 1024), (514, 3072, 1024), (77, 64, 640)]


@pytest.mark.parametrize("M,N,K", LIN_SHAPES)
@pytest.mark.parametrize("act", [0, 1, 2])
def test_linear_f32(M, N, K, act):
    a, w, b, r = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5), _rand(N, seed=3), _rand(M, N, seed=4)
    ref = F.linear(a.double(), w.double(), b.double())
    ref = [ref, O.quick_gelu(ref), F.gelu(ref)][act] + r.double()
    got = ops.linear(a.to(DEV), w.to(DEV), b.to(DEV), r.to(DEV), act=act)
    assert _rel_err(got, ref) < 2e-6              # exact-f32 MFMA: fp32 rounding class only


@pytest.mark.parametrize("M,N,K", LIN_SHAPES)
@pytest.mark.parametrize("act", [0, 1, 2])
def test_linear_bf16(M, N, K, act):
    a, w = _rand(M, K, seed=1).bfloat16(), (_rand(N, K, seed=2, scale=K ** -0.5)).bfloat16()
    b, r = _rand(N, seed=3), _rand(M, N, seed=4).bfloat16()
    ref = F.linear(a.double(), w.double(), b.double())
    ref = [ref, O.quick_gelu(ref), F.gelu(ref)][act] + r.double()
    got = ops.linear(a.to(DEV), w.to(DEV), b.to(DEV), r.to(DEV), act=act)
    assert got.dtype == torch.bfloat16
    assert _rel_err(got, ref) < 6e-3              # one bf16 rounding of the result (2^-8) + fp32 accumulation
    got32 = ops.linear(a.to(DEV), w.to(DEV), b.to(DEV), r.float().to(DEV), act=act, out_dtype=torch.float32)
    assert _rel_err(got32, ref) < 2e-5            # bf16 products are exact in fp32; only the accumulation order differs


@pytest.mark.parametrize("M,N,K,act,res", [(4096, 4096, 1024, 1, True), (4100, 2056, 128, 0, False), (2560, 3072, 256, 2, True),
                                           (6000, 1024, 4096, 0, True), (8448, 2048, 512, 1, True), (65792, 1024, 128, 0, True),
                                           (16640, 4096, 64, 2, False), (4100, 2112, 192, 0, True), (6000, 1024, 256, 1, False),
                                           (25700, 1344, 320, 2, True), (66000, 768, 768, 0, True),
                                           # round 5: every activation x residual combination on the ping-pong kernel (N a multiple of 256, K >= 128) —
                                           # the erf-GELU form without a residual had no case here, and it is the one whose epilogue reuses the store
                                           # registers at once (the streaming stores' wait states, gemm_persist.hip)
                                           (7776, 3072, 768, 2, False), (8192, 4096, 1024, 2, False), (7680, 768, 3072, 1, True), (7776, 768, 768, 2, True),
                                           (23040, 2304, 768, 1, False), (7680, 1024, 1024, 0, False)])
def test_linear_bf16_large_tiles(M, N, K, act, res):
    """Shapes with >= 96 output tiles take the 256x256 direct-to-LDS kernel (ragged M and N edges included)."""
    a, w = _rand(M, K, seed=1).bfloat16(), (_rand(N, K, seed=2, scale=K ** -0.5)).bfloat16()
    b = _rand(N, seed=3)
    r = _rand(M, N, seed=4).bfloat16() if res else None
    ref = F.linear(a.double(), w.double(), b.double())
    ref = [ref, O.quick_gelu(ref), F.gelu(ref)][act]
    if res:
        ref = ref.bfloat16().double() + r.double()         # torch-bf16 semantics: the Linear output is rounded, then added
    got = ops.linear(a.to(DEV), w.to(DEV), b.to(DEV), None if r is None else r.to(DEV), act=act)
    assert _rel_err(got, ref) < 6e-3
    # element-wise: at most one bf16 ulp (2^-8 relative) + accumulation-order noise
    err = (got.double().cpu() - ref).abs()
    assert bool((err <= 2.0 ** -7 * ref.abs() + 2e-2).all())


def test_linear_bf16_large_no_bias_and_alias():
    """Persistent kernel without a bias (scalar-loaded zero row) and with C aliasing the residual (in-place residual stream)."""
    M, N, K = 8192, 3072, 1024
    a, w = _rand(M, K, seed=1).bfloat16(), (_rand(N, K, seed=2, scale=K ** -0.5)).bfloat16()
    ref = F.linear(a.double(), w.double())
    got = ops.linear(a.to(DEV), w.to(DEV))
    assert _rel_err(got, ref) < 6e-3
    r = _rand(M, N, seed=4).bfloat16()
    buf = r.to(DEV).clone()
    ops.linear(a.to(DEV), w.to(DEV), None, residual=buf, out=buf)
    assert _rel_err(buf, ref.bfloat16().double() + r.double()) < 6e-3


@pytest.mark.parametrize("N,K", [(1024, 1024), (4096, 1024), (1024, 4096)])
@pytest.mark.parametrize("act,use_res", [(0, False), (1, False), (2, True), (0, True)])
def test_linear_bf16_rows_do_not_depend_on_the_kernel(act, use_res, N, K):
    """The same rows through the persistent 256 x 256 kernel (M = 25 k: 392 tiles), through its small-problem kernel in all three shapes
    (64 x 32 tiles where 64 x 64 ones would leave half the chip idle: M = 1, 63, 257 at N = 1024; four stages / two workgroups per CU where
    there are more tiles than CUs: M = 1028 at N = 1024, M = 257 at N = 4096; 64 x 64 with eight stages otherwise) and inside a mid-size
    problem give the same bits: a sample alone equals the sample inside a batch."""
    M = 25088
    a, w = _rand(M, K, seed=11).bfloat16().to(DEV), (_rand(N, K, seed=12, scale=K ** -0.5)).bfloat16().to(DEV)
    b = _rand(N, seed=13).to(DEV)
    r = _rand(M, N, seed=14).bfloat16().to(DEV) if use_res else None
    big = ops.linear(a, w, b, r, act=act)
    for m in (1, 63, 257, 1028, 2056, 2048, 4112):          # 2056 / 2048 at N = 1024: 128 x 64 tiles in one round + 8 / 0 rows behind them inside the launch (round 5)
        small = ops.linear(a[:m].contiguous(), w, b, None if r is None else r[:m].contiguous(), act=act)
        assert torch.equal(small, big[:m]), m


def test_linear_transpose_detecting():
    """A = I with an asymmetric W catches a swapped C/D fragment mapping (cdna guide rule 16)."""
    K = 128
    w = torch.arange(K * K, dtype=torch.float32).reshape(K, K) / (K * K)
    for dt, tol in ((torch.float32, 1e-7), (torch.bfloat16, 4e-3)):
        got = ops.linear(torch.eye(K).to(dt).to(DEV), w.to(dt).to(DEV))
        assert _rel_err(got, w.to(dt).double().t()) < tol


# ---------------------------------------------------------------------------------------------
# layernorm / attention
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-6), (torch.bfloat16, 8e-3), (torch.float16, 8e-3)])
@pytest.mark.parametrize("rows,C", [(5, 64), (1028, 1024), (33, 768), (7776, 768), (4500, 1280), (1030, 512), (3100, 2048), (5000, 1024), (3000, 1536)])
def test_layernorm(dt, tol, rows, C):
    x = (_rand(rows, C, seed=5) * 3 + 0.5).to(dt)
    g, b = 1 + 0.1 * _rand(C, seed=6), 0.1 * _rand(C, seed=7)
    ref = F.layer_norm(x.double(), (C,), g.double(), b.double(), 1e-5)
    got = ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), 1e-5)
    assert _rel_err(got, ref) < tol
    if dt == torch.bfloat16 and rows >= 1024:
        # the register-resident rows kernel (>= 1024 rows of a width it knows, incl. 768 / 1280 whose last 64-lane chunk is partly empty) and the
        # generic kernel (fewer rows) share their arithmetic: a row alone equals the row inside the batch
        xd = x.to(DEV)
        parts = torch.cat([ops.layernorm(xd[i:i + 500].contiguous(), g.to(DEV), b.to(DEV), 1e-5) for i in range(0, rows, 500)], 0)
        assert torch.equal(parts, got)                                     # EVERY row (10 of 7776 differed in one element before the fma's were pinned)
        buf = xd.clone()
        assert torch.equal(ops.layernorm(buf, g.to(DEV), b.to(DEV), 1e-5, out=buf), got)      # in place, as the decoder runs it


def _attn_ref(qkv, H, Dh, scale, offsets):
    qkv = qkv.double()
    out = torch.zeros(qkv.shape[0], H * Dh, dtype=torch.float64)
    for s0, s1 in zip(offsets[:-1], offsets[1:]):
        blk = qkv[s0:s1].reshape(s1 - s0, 3, H, Dh).permute(1, 2, 0, 3)
        att = torch.softmax(blk[0] @ blk[1].transpose(-1, -2) * scale, dim=-1)
        out[s0:s1] = (att @ blk[2]).transpose(0, 1).reshape(s1 - s0, H * Dh)
    return out


@pytest.mark.parametrize("dt,tol", [(torch.float32, 3e-6), (torch.bfloat16, 1e-2), (torch.float16, 1e-2)])
@pytest.mark.parametrize("T,H,Dh,nimg", [(17, 4, 16, 3), (257, 16, 64, 2), (65, 2, 512, 2), (197, 12, 64, 1), (324, 16, 48, 2), (33, 3, 48, 3),
                                          (256, 12, 64, 2), (577, 16, 64, 1), (40, 2, 96, 2),
                                          (300, 16, 64, 1), (1, 2, 64, 2), (608, 4, 64, 1), (289, 16, 48, 1), (9, 2, 64, 2), (41, 2, 64, 2)])
def test_attention_uniform(dt, tol, T, H, Dh, nimg):
    qkv = _rand(nimg * T, 3 * H * Dh, seed=8).to(dt)
    ref = _attn_ref(qkv, H, Dh, Dh ** -0.5, [i * T for i in range(nimg + 1)])
    got = ops.attention(qkv.to(DEV), H, Dh, Dh ** -0.5, seg_len=T)
    assert _rel_err(got, ref) < tol


@pytest.mark.parametrize("dt,tol", [(torch.float32, 3e-6), (torch.bfloat16, 1e-2), (torch.float16, 1e-2)])
def test_attention_ragged(dt, tol):
    H, Dh = 2, 32
    lens = [1, 7, 64, 3, 1, 129, 20]
    offs = np.concatenate([[0], np.cumsum(lens)]).tolist()
    qkv = _rand(offs[-1], 3 * H * Dh, seed=9).to(dt)
    ref = _attn_ref(qkv, H, Dh, Dh ** -0.5, offs)
    so = torch.tensor(offs, dtype=torch.int32, device=DEV)
    got = ops.attention(qkv.to(DEV), H, Dh, Dh ** -0.5, seg_len=max(lens), seg_offsets=so, n_segs=len(lens))
    assert _rel_err(got, ref) < tol


def _cross_ref(q, k, v, H, Dh, scale, q_len, offs):
    """fp64 reference of the padded reference arithmetic: per segment, softmax(q k^T * scale) v over ITS keys."""
    out = torch.zeros(q.shape[0], H * Dh, dtype=torch.float64)
    for s in range(len(offs) - 1):
        qs = q[s * q_len:(s + 1) * q_len].double().reshape(q_len, H, Dh).transpose(0, 1)
        ks = k[offs[s]:offs[s + 1]].double().reshape(-1, H, Dh).transpose(0, 1)
        vs = v[offs[s]:offs[s + 1]].double().reshape(-1, H, Dh).transpose(0, 1)
        a = torch.softmax(qs @ ks.transpose(-1, -2) * scale, -1)
        out[s * q_len:(s + 1) * q_len] = (a @ vs).transpose(0, 1).reshape(q_len, H * Dh)
    return out


@pytest.mark.parametrize("dt,tol", [(torch.float32, 3e-6), (torch.bfloat16, 1e-2), (torch.float16, 1e-2)])
@pytest.mark.parametrize("H,Dh,q_len,lens", [(4, 16, 25, [7, 3, 1, 12]), (12, 64, 256, [37, 24, 1, 64, 65]), (12, 64, 324, [40, 33]),
                                             (2, 64, 5, [130, 2]), (3, 96, 16, [9, 70]), (12, 64, 33, [50, 17, 1]), (4, 64, 257, [8, 9, 31, 32, 33])])
def test_cross_attention_ragged(dt, tol, H, Dh, q_len, lens):
    """setok_cross_attention (module.py:283-286,303,342-364): query groups x ragged key segments, k / v as column windows of one
    fused buffer; Dh = 64 in bf16 takes the MFMA kernel, everything else the generic one."""
    offs = np.concatenate([[0], np.cumsum(lens)]).tolist()
    C = H * Dh
    q = _rand(len(lens) * q_len, C, seed=21).to(dt)
    kv = _rand(offs[-1], 2 * C, seed=22).to(dt)
    ref = _cross_ref(q, kv[:, :C], kv[:, C:], H, Dh, Dh ** -0.5, q_len, offs)
    kvd = kv.to(DEV)
    so = torch.tensor(offs, dtype=torch.int32, device=DEV)
    got = ops.cross_attention(q.to(DEV), kvd[:, :C], kvd[:, C:], H, Dh, Dh ** -0.5, q_len, so, len(lens), max(lens))
    assert _rel_err(got, ref) < tol
    # the mask of the padded reference is equivalent: additive -10000 on the padded keys (module.py:849)
    if dt == torch.float32:
        L = max(lens)
        kp = torch.zeros(len(lens), L, 2 * C); add = torch.full((len(lens), 1, 1, L), -10000.0)
        for s, n in enumerate(lens):
            kp[s, :n] = kv[offs[s]:offs[s + 1]]; add[s, ..., :n] = 0
        qs = q.reshape(len(lens), q_len, H, Dh).transpose(1, 2)
        ks = kp[..., :C].reshape(len(lens), L, H, Dh).transpose(1, 2); vs = kp[..., C:].reshape(len(lens), L, H, Dh).transpose(1, 2)
        padded = (torch.softmax(qs @ ks.transpose(-1, -2) * Dh ** -0.5 + add, -1) @ vs).transpose(1, 2).reshape(-1, C)
        assert _rel_err(got, padded) < 1e-5


def test_cross_attention_uniform_and_errors():
    H, Dh, q_len, L, B = 2, 32, 6, 5, 3
    C = H * Dh
    q, kv = _rand(B * q_len, C, seed=23), _rand(B * L, 2 * C, seed=24)
    ref = _cross_ref(q, kv[:, :C], kv[:, C:], H, Dh, 0.3, q_len, [0, 5, 10, 15])
    kvd = kv.to(DEV)
    got = ops.cross_attention(q.to(DEV), kvd[:, :C], kvd[:, C:], H, Dh, 0.3, q_len, None, B, L)
    assert _rel_err(got, ref) < 3e-6
    with pytest.raises(Exception):
        ops.cross_attention(q.to(DEV), kvd[:, :C], kvd[:, C:], H, Dh, 0.3, 0, None, B, L)


@pytest.mark.parametrize("lens", [[1, 7, 64, 3, 1, 129, 20], [5] * 40, [33, 32, 31, 1, 2, 300]])
def test_attention_ragged_head_dim_512_bf16(lens):
    """The SeTok head's shape (2 heads x 512) over ragged segments: the segment-owning MFMA kernel (attn_seg.hip)."""
    H, Dh = 2, 512
    offs = np.concatenate([[0], np.cumsum(lens)]).tolist()
    qkv = (_rand(offs[-1], 3 * H * Dh, seed=31) * 0.5).bfloat16()
    ref = _attn_ref(qkv, H, Dh, Dh ** -0.5, offs)
    so = torch.tensor(offs, dtype=torch.int32, device=DEV)
    got = ops.attention(qkv.to(DEV), H, Dh, Dh ** -0.5, seg_len=max(lens), seg_offsets=so, n_segs=len(lens))
    assert _rel_err(got, ref) < 1e-2
    again = ops.attention(qkv.to(DEV), H, Dh, Dh ** -0.5, seg_len=max(lens), seg_offsets=so, n_segs=len(lens))
    assert torch.equal(got, again)


# ---------------------------------------------------------------------------------------------
# glue
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,p,g", [(2, 14, 4), (3, 16, 14), (1, 32, 7), (5, 2, 3), (2, 7, 5), (1, 3, 2)])
def test_patchify_matches_conv(dt, B, p, g):
    img = _rand(B, 3, p * g, p * g, seed=10).to(dt)
    w = _rand(8, 3, p, p, seed=11).to(dt)
    kpad = ops.round_up(3 * p * p, 64)
    pat = ops.patchify(img.to(DEV), p, kpad).cpu()
    assert pat.shape == (B * g * g, kpad) and (kpad == 3 * p * p or float(pat[:, 3 * p * p:].abs().max()) == 0.0)
    ref = F.conv2d(img.double(), w.double(), stride=p).flatten(2).transpose(1, 2).reshape(B * g * g, 8)
    got = pat[:, :3 * p * p].double() @ w.double().reshape(8, -1).t()
    assert _rel_err(got, ref) < 1e-12


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
def test_assemble_and_select(dt):
    B, N, C = 3, 16, 64
    pe, cls, pos = _rand(B * N, C, seed=12).to(dt), _rand(C, seed=13).to(dt), _rand(N + 1, C, seed=14).to(dt)
    tok = ops.vit_assemble(pe.to(DEV), cls.to(DEV), pos.to(DEV), B, N).cpu().reshape(B, N + 1, C)
    ref = (torch.cat([cls.float().expand(B, 1, C), pe.float().reshape(B, N, C)], 1) + pos.float()[None]).to(dt)
    assert torch.equal(tok, ref)
    p2 = O.pos_encoding_2d(4, 4, C, dt)
    for skip in (1, 0):
        hid = _rand(B, N + skip, C, seed=15).to(dt)
        x = ops.select_add_pos(hid.to(DEV), p2.to(DEV), B, N, skip).cpu().reshape(B, N, C)
        assert torch.equal(x, (hid[:, skip:].float() + p2.float()[None]).to(dt))


# ---------------------------------------------------------------------------------------------
# clustering against the reference's golden vectors (integers bit-exact)
# ---------------------------------------------------------------------------------------------
def _cluster_gpu(x, k, thr, mcn, noise=None, token_mask=None):
    N = x.shape[0]
    idx, score, index_down, counts = ops.cluster_dpc_knn(
        x.to(DEV), 1, N, k, thr, mcn,
        None if noise is None else noise.reshape(1, N), None if token_mask is None else token_mask.reshape(1, N))
    L = int(counts[0])
    return idx[0].cpu(), score.cpu(), index_down[0, :L].cpu(), L, index_down[0, L:].cpu()


def test_cluster_head_small_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "head_small.npz"))
    for case in ("fallback", "dynamic", "planted", "masked", "k_explicit"):
        x = torch.from_numpy(z[f"{case}:x"])
        k = int(z[f"{case}:k"]); thr = float(z[f"{case}:threshold"])
        k = 8 if k < 0 else k
        thr = 0.5 if thr < 0 else thr
        nz = torch.from_numpy(z[f"{case}:noise"]) if f"{case}:noise" in z.files else None
        tm = torch.from_numpy(z[f"{case}:token_mask"]) if f"{case}:token_mask" in z.files else None
        idx, score, centres, L, pad = _cluster_gpu(x, k, thr, 8, nz, tm)
        sens = O.cluster_sensitivity(x, k, thr, 8, tm, nz)
        assert sens["centres_certain"] and bool(sens["assign_certain"].all()), case      # fixtures chosen with wide margins
        O.check_cluster_parity(centres, idx, torch.from_numpy(z[f"{case}:index_down"]), torch.from_numpy(z[f"{case}:idx_cluster"]), sens)
        assert torch.equal(centres, torch.from_numpy(z[f"{case}:index_down"])), case
        assert torch.equal(idx, torch.from_numpy(z[f"{case}:idx_cluster"])), case
        assert bool((pad == -1).all())
        O.check_score(score, sens)


def test_cluster_full_dims_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "cluster_full.npz"))
    n_exact = 0
    for name in sorted({k.split(":")[0] for k in z.files}):
        N, C, m, seed, k, mcn, thr = z[name + ":spec"]
        N, C, m, seed, k, mcn = int(N), int(C), int(m), int(seed), int(k), int(mcn)
        h = int(N ** 0.5)
        x = O.planted_features(N, C, m, seed=seed) + O.pos_encoding_2d(h, h, C)
        idx, score, centres, L, _ = _cluster_gpu(x, k, float(thr), mcn)
        sens = O.cluster_sensitivity(x, k, float(thr), mcn)
        st = O.check_cluster_parity(centres, idx, torch.from_numpy(z[name + ":index_down"]).long(),
                                    torch.from_numpy(z[name + ":idx_cluster"]).long(), sens)
        n_exact += st["centres_certain"]
        if st["centres_certain"]:
            assert st["tokens_certain"] == N and st.get("tokens_equal") == N, (name, st)
        O.check_score(score, sens)
    assert n_exact >= 10        # only the two top-32 fallback cases sit on a rounding-level score tie (sensitivity analysis)


def test_cluster_vitl_reference_features(golden_dir):
    """Dynamic-k and fallback branches on the reference's own ViT-L tower features (cfg2 dims)."""
    z = np.load(os.path.join(golden_dir, "vitl_224.npz"))
    feats = torch.from_numpy(z["feats"])
    x = feats + O.pos_encoding_2d(16, 16, 1024)[None]
    for thr, pre in ((0.125, ""), (0.5, "fb:")):
        idx, score, index_down, counts = ops.cluster_dpc_knn(x.to(DEV).reshape(-1, 1024), 2, 256, 64, thr, 64)
        for i in range(2):
            L = int(counts[i])
            sens = O.cluster_sensitivity(x[i], 64, thr, 64)
            assert sens["centres_certain"] and bool(sens["assign_certain"].all())
            ref_c = torch.from_numpy(z[f"{i}:{pre}index_down"]).long()
            ref_i = torch.from_numpy(z[f"{i}:{pre}idx_cluster"]).long()
            st = O.check_cluster_parity(index_down[i, :L].cpu(), idx[i].cpu(), ref_c, ref_i, sens)
            assert st["tokens_equal"] == 256
            O.check_score(score[i].cpu(), sens)


def test_cluster_batched_equals_per_image_and_bf16_runs():
    B, N, C = 5, 256, 1024
    xs = torch.stack([O.planted_features(N, C, 3 + i, seed=50 + i) for i in range(B)])
    idx, score, index_down, counts = ops.cluster_dpc_knn(xs.to(DEV).reshape(-1, C), B, N, 8, 0.5, 64)
    for i in range(B):
        r = O.cluster_dpc_knn(xs[i], 8, 0.5, 64)
        O.check_cluster_parity(index_down[i, :int(counts[i])].cpu(), idx[i].cpu(), r.index_down, r.idx_cluster,
                               O.cluster_sensitivity(xs[i], 8, 0.5, 64))
    # bf16 inputs: same algorithm on the bf16-rounded features (exact products, fp32 accumulate)
    xb = xs.bfloat16()
    idx_b, _, down_b, counts_b = ops.cluster_dpc_knn(xb.to(DEV).reshape(-1, C), B, N, 8, 0.5, 64)
    for i in range(B):
        r = O.cluster_dpc_knn(xb[i].float(), 8, 0.5, 64)
        O.check_cluster_parity(down_b[i, :int(counts_b[i])].cpu(), idx_b[i].cpu(), r.index_down, r.idx_cluster,
                               O.cluster_sensitivity(xb[i].float(), 8, 0.5, 64))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,C,k,mcn,thr", [(1024, 256, 64, 64, 0.5), (1024, 256, 64, 64, 1e9), (4, 64, 2, 2, 0.5), (1, 64, 1, 1, 0.5), (576, 1024, 64, 64, 0.2)])
def test_cluster_size_limits(dt, N, C, k, mcn, thr):
    """The largest supported grid (32 x 32 = 1024 patches: 64 lanes x 16 register slots per distance row), the smallest (2 x 2, and a single
    patch), and the 336^2 grid, fp32 and bf16 (bf16: the dedicated Gram kernel incl. its column-chunk loop for N > 256), against the oracle on
    the same (rounded) features; decisions the fp64 margin analysis calls certain must be equal."""
    xs = torch.stack([O.planted_features(N, C, max(1, min(N, 5)) + i, seed=70 + i) for i in range(2)]).to(dt)
    idx, score, index_down, counts = ops.cluster_dpc_knn(xs.to(DEV).reshape(-1, C), 2, N, k, thr, mcn)
    for i in range(2):
        r = O.cluster_dpc_knn(xs[i].float(), k, thr, mcn)
        L = int(counts[i])
        assert idx.dtype == torch.int64 and int(idx[i].max()) < L and int(idx[i].min()) >= 0
        O.check_cluster_parity(index_down[i, :L].cpu(), idx[i].cpu(), r.index_down, r.idx_cluster,
                               O.cluster_sensitivity(xs[i].float(), k, thr, mcn))


def test_cluster_argument_errors():
    x = torch.zeros(1025, 64, device=DEV)
    with pytest.raises(Exception):
        ops.cluster_dpc_knn(x, 1, 1025, 8, 0.5, 8)                 # N > 1024
    with pytest.raises(Exception):
        ops.cluster_dpc_knn(x[:16], 1, 16, 17, 0.5, 8)             # k > N: torch.topk would raise in the reference
    with pytest.raises(Exception):
        ops.cluster_dpc_knn(x[:16], 1, 16, 4, 0.5, 17)             # min_cluster_num > N


# ---------------------------------------------------------------------------------------------
# sort / gather / segment mean
# ---------------------------------------------------------------------------------------------
def test_sort_gather_segment_mean():
    B, N, C = 4, 64, 64
    g = torch.Generator().manual_seed(20)
    Ls = [1, 5, 64, 17]
    idx = torch.stack([torch.cat([torch.arange(L), torch.randint(0, L, (N - L,), generator=g)])[torch.randperm(N, generator=g)]
                       for L in Ls])
    counts = torch.tensor(Ls, dtype=torch.int32)
    perm, seg, img = ops.cluster_sort(idx.to(DEV), counts.to(DEV))
    total = sum(Ls)
    assert img.cpu().tolist() == np.concatenate([[0], np.cumsum(Ls)]).tolist()
    perm, seg = perm.cpu().long(), seg.cpu()[:total + 1].long()
    exp_perm, exp_seg = [], []
    for b in range(B):
        order = torch.sort(idx[b], stable=True).indices
        exp_perm.append(order + b * N)
        sizes = torch.bincount(idx[b], minlength=Ls[b])
        exp_seg.append(b * N + torch.cumsum(sizes, 0) - sizes)
    assert torch.equal(perm, torch.cat(exp_perm))
    assert torch.equal(seg, torch.cat(exp_seg + [torch.tensor([B * N])]))
    for dt, tol in ((torch.float32, 1e-6), (torch.bfloat16, 8e-3)):
        x = _rand(B * N, C, seed=21).to(dt)
        xs = ops.gather_rows(x.to(DEV), perm.int().to(DEV))
        assert torch.equal(xs.cpu(), x[perm])
        mean = ops.segment_mean(xs, seg.int().to(DEV), img.to(DEV)[B:], total).cpu()
        ref = torch.stack([x[perm][seg[s]:seg[s + 1]].double().mean(0) for s in range(total)])
        assert _rel_err(mean, ref) < tol


# ---------------------------------------------------------------------------------------------
# the single-launch, register-resident form of the clustering (bf16, N <= 256): every branch against the oracle on the same bf16-valued
# features, and against the multi-kernel form of the same library (SETOK_CLUSTER_FUSED=0)
# ---------------------------------------------------------------------------------------------
FUSED_CASES = [
    # N, C, planted regions, k, mcn, threshold, noise?, mask?
    (256, 1024, 6, 64, 64, 0.5, False, False),      # cfg2 shape, dynamic-k on planted features
    (256, 1024, 6, 64, 64, 1e9, False, False),      # fallback: the 64 best scores
    (256, 1024, 9, 8, 64, 0.5, True, False),        # tie-break noise (tokenizer.py:91)
    (256, 1024, 5, 8, 16, 0.5, False, True),        # token_mask (:84-86, :93-94)
    (256, 1024, 5, 8, 16, 1e9, True, True),         # mask + noise + fallback
    (196, 768, 7, 32, 32, 0.5, False, False),       # 14 x 14 grid (ViT-B/16), 768 channels: rows beyond N are padding inside the kernel
    (196, 768, 7, 32, 32, 0.5, False, True),
    (64, 64, 4, 8, 8, 0.5, False, False),           # one K-tile
    (64, 128, 4, 48, 8, 0.5, True, False),
    (64, 128, 4, 64, 8, 0.5, False, False),         # k == N: every density is the mean over the whole row
    (16, 64, 3, 4, 4, 0.5, False, True),
    (4, 64, 2, 2, 2, 0.5, False, False),
    (1, 64, 1, 1, 1, 0.5, False, False),
]


@pytest.mark.parametrize("N,C,m,k,mcn,thr,with_noise,with_mask", FUSED_CASES)
def test_cluster_fused_bf16_every_branch(N, C, m, k, mcn, thr, with_noise, with_mask):
    B = 3
    g = torch.Generator().manual_seed(N * 31 + C + k)
    xs = torch.stack([O.planted_features(N, C, max(1, min(N, m)) + i, seed=90 + i) for i in range(B)]).bfloat16()
    noise = torch.rand(B, N, generator=g) if with_noise else None
    mask = None
    if with_mask:
        mask = (torch.rand(B, N, generator=g) > 0.3).float()
        mask[:, 0] = 1.0
    assert os.environ.get("SETOK_CLUSTER_FUSED") is None
    idx, score, index_down, counts = ops.cluster_dpc_knn(xs.to(DEV).reshape(-1, C), B, N, k, thr, mcn, noise, mask)
    os.environ["SETOK_CLUSTER_FUSED"] = "0"
    try:
        idx_m, score_m, down_m, counts_m = ops.cluster_dpc_knn(xs.to(DEV).reshape(-1, C), B, N, k, thr, mcn, noise, mask)
    finally:
        del os.environ["SETOK_CLUSTER_FUSED"]
    n_same = 0
    for i in range(B):
        nz = None if noise is None else noise[i]
        tm = None if mask is None else mask[i]
        r = O.cluster_dpc_knn(xs[i].float(), k, thr, mcn, tm, nz)
        # k == N: a density is the mean over the WHOLE row, tokens of one planted region tie to ~1e-7 and rounding of exp / the mean decides
        # their order — beyond what a per-entry perturbation of d^2 models; widen it there
        sens = O.cluster_sensitivity(xs[i].float(), k, thr, mcn, tm, nz, ulps=4.0 if k < N else 4096.0)
        # the score envelope: the MFMA sums the C products of a Gram entry in its own order, sqrt(C) * 2^-24 |a||b| away from the CPU's order
        # (tens of ulps of |a|^2 + |b|^2 at C = 1024), so near-tied densities of neighbouring planted tokens may order differently
        sens_score = O.cluster_sensitivity(xs[i].float(), k, thr, mcn, tm, nz, ulps=64.0 if k < N else 4096.0)
        L = int(counts[i])
        assert int(idx[i].max()) < L and int(idx[i].min()) >= 0 and bool((index_down[i, L:] == -1).all())
        try:
            O.check_cluster_parity(index_down[i, :L].cpu(), idx[i].cpu(), r.index_down, r.idx_cluster, sens)
        except AssertionError as ex:
            Lm = int(counts_m[i])
            sel = set(index_down[i, :L].cpu().tolist()); ref_sel = set(r.index_down.tolist()); m_sel = set(down_m[i, :Lm].cpu().tolist())
            extra = sorted(sel - ref_sel)[:4]
            gs, ms, rs = score[i].cpu().double(), score_m[i].cpu().double(), r.score.reshape(-1).double()
            raise AssertionError(f"{ex}; image {i}: fused L={L} multi L={Lm} oracle L={len(ref_sel)}; fused-only centres {extra}; multi == oracle: {m_sel == ref_sel}; "
                                 + "; ".join(f"tok {t}: score fused {gs[t]:.6g} multi {ms[t]:.6g} oracle {rs[t]:.6g} rho {float(r.density[t]):.7g} delta {float(r.delta[t]):.6g}"
                                             for t in extra))
        try:
            O.check_score(score[i].cpu(), sens_score)
        except AssertionError as ex:
            sens = sens_score
            gs, ms, rs = score[i].cpu().double(), score_m[i].cpu().double(), r.score.reshape(-1).double()
            bad = ((gs < sens["score_lo"] * (1 - 1e-3) - 1e-7) | (gs > sens["score_hi"] * (1 + 1e-3) + 1e-7)).nonzero().reshape(-1).tolist()
            raise AssertionError(f"{ex}; image {i}; " + "; ".join(
                f"tok {t}: fused {gs[t]:.6g} multi {ms[t]:.6g} oracle {rs[t]:.6g} env [{sens['score_lo'][t]:.6g}, {sens['score_hi'][t]:.6g}] "
                f"mask {None if tm is None else float(tm[t])} rho {float(r.density[t]):.6g} delta {float(r.delta[t]):.6g}" for t in bad[:6]))
        Lm = int(counts_m[i])
        O.check_cluster_parity(down_m[i, :Lm].cpu(), idx_m[i].cpu(), r.index_down, r.idx_cluster, sens)
        n_same += int(torch.equal(idx[i], idx_m[i]) and L == Lm)
    print(f"fused vs multi-kernel: {n_same}/{B} images with identical integers")


@pytest.mark.parametrize("seed", list(range(24)))
def test_cluster_fused_random_shapes_against_the_oracle(seed):
    """Seeded random (N, C, k, min_cluster_num, threshold) with the data the shortcuts of the fused kernel care about — exact duplicate tokens
    (distances of exactly 0 beside the self-distance: no common prefix for the radix select), N < 256 (columns padded with +inf), k = 2 and
    k = N — against the fp64 margin analysis of the oracle (the contract of the bf16 mode: decisions equal wherever 64 ulps of d^2 cannot flip
    them, scores inside the envelope), and against the multi-kernel form of the same call (equal selections).
    Left out on purpose: k = 1 and constant features.  With k = 1 every density is exp(-d_ii^2); the kernels' self-distance is exactly 0 (the
    norms ARE the Gram diagonal) while torch.cdist's matmul form leaves rounding noise there (tokenizer.py:82), so the reference's own result is
    decided by that noise; with identical tokens every score ties at 0 and torch.topk's order among ties is unspecified."""
    import random
    rng = random.Random(1000 + seed)
    N = rng.choice([256, 256, 200, 129, 64, 33, 16, rng.randint(2, 256)])
    C = 64 * rng.randint(1, 8)
    k = rng.choice([2, N, min(N, 64), rng.randint(2, N)])
    mcn = rng.randint(1, min(N, 64))
    thr = rng.choice([0.5, 0.1, 1e9])
    m = rng.randint(2, 12)
    gx = torch.Generator().manual_seed(500 + seed)                # m planted centres, any N (O.planted_features wants a square grid)
    x = (torch.randn(min(N, m), C, generator=gx) * 2.0)[torch.randint(0, min(N, m), (N,), generator=gx)] + 0.05 * torch.randn(N, C, generator=gx)
    dups = seed % 3 == 0
    if dups:                                             # exact duplicates
        for _ in range(rng.randint(1, 6)):
            a, b = rng.randrange(N), rng.randrange(N)
            x[a] = x[b]
    x = x.bfloat16()
    g = torch.Generator().manual_seed(seed)
    noise = torch.rand(N, generator=g) if seed % 2 else None
    nz = None if noise is None else noise[None].to(DEV)
    idx, score, index_down, counts = ops.cluster_dpc_knn(x.to(DEV), 1, N, k, thr, mcn, nz, None)
    os.environ["SETOK_CLUSTER_FUSED"] = "0"
    try:
        idx_m, score_m, down_m, counts_m = ops.cluster_dpc_knn(x.to(DEV), 1, N, k, thr, mcn, nz, None)
    finally:
        del os.environ["SETOK_CLUSTER_FUSED"]
    L = int(counts[0])
    assert L == int(counts_m[0]) and torch.equal(index_down[0, :L], down_m[0, :L]) and torch.equal(idx, idx_m)       # the two forms select alike
    r = O.cluster_dpc_knn(x.float(), k, thr, mcn, None, noise)
    wide = k == N or dups                                # whole-row means / exact ties: rounding of exp and of the mean decides, beyond a d^2 perturbation
    sens = O.cluster_sensitivity(x.float(), k, thr, mcn, None, noise, ulps=4096.0 if wide else 64.0)
    assert 1 <= L <= N and int(idx[0].max()) < L and int(idx[0].min()) >= 0 and bool(torch.isfinite(score).all())
    O.check_cluster_parity(index_down[0, :L].cpu(), idx[0].cpu(), r.index_down, r.idx_cluster, sens)
    if not wide:
        O.check_score(score[0].cpu(), sens)


def test_cluster_fused_needs_no_workspace_and_is_deterministic():
    from setok_amd import _lib
    import ctypes
    nd, nv = ctypes.c_int64(-1), ctypes.c_int64(-1)
    _lib.call("setok_cluster_workspace", ops.BF16, 256, 256, 1024, ctypes.byref(nd), ctypes.byref(nv))
    assert nd.value == 0 and nv.value == 0                               # bf16, N <= 256: one launch, no distance matrix in memory
    _lib.call("setok_cluster_workspace", ops.BF16, 128, 576, 1024, ctypes.byref(nd), ctypes.byref(nv))
    assert nd.value == 0 and 0 < nv.value < 128 * 4 * 576 * 2            # 256 < N <= 576: one launch over 128-row strips, no N x N matrix either; a few vectors per image
    _lib.call("setok_cluster_workspace", ops.BF16, 4, 729, 1152, ctypes.byref(nd), ctypes.byref(nv))
    assert nd.value == 4 * 729 * 729                                        # N > 576 (27 x 27 patches): the multi-kernel path and its matrix
    _lib.call("setok_cluster_workspace", ops.F32, 2, 256, 1024, ctypes.byref(nd), ctypes.byref(nv))
    assert nd.value == 2 * 256 * 256
    xs = torch.stack([O.planted_features(256, 1024, 5 + i, seed=120 + i) for i in range(64)]).bfloat16().to(DEV).reshape(-1, 1024)
    a = ops.cluster_dpc_knn(xs, 64, 256, 64, 0.5, 64)
    b = ops.cluster_dpc_knn(xs, 64, 256, 64, 0.5, 64)
    assert all(torch.equal(u, v) for u, v in zip(a, b))
    one = ops.cluster_dpc_knn(xs[5 * 256: 6 * 256], 1, 256, 64, 0.5, 64)
    assert torch.equal(one[0][0], a[0][5]) and torch.equal(one[1][0], a[1][5]) and int(one[3][0]) == int(a[3][5])     # image alone == in the batch


# ---------------------------------------------------------------------------------------------
# LayerNorm folded into the consuming Linear (bf16 throughput mode)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K,act", [(65792 // 8, 3072, 1024, 0), (65792 // 8, 4096, 1024, 1), (771, 3072, 1024, 0), (257, 4096, 1024, 1),
                                       (300, 128, 64, 2), (5, 64, 192, 0), (8224, 1024, 1024, 2)])
def test_linear_ln_equals_layernorm_then_linear(M, N, K, act):
    """setok_row_stats + setok_ln_fold + setok_linear_ln against fp64 LayerNorm -> Linear -> activation on the same bf16 inputs, with a
    mean and scale per row that make the rank-1 correction matter; and against the unfolded bf16 pipeline (separate setok_layernorm): the
    folded form skips one bf16 rounding of the normalised activations, so it must not be further from fp64 than the unfolded one (+ margin)."""
    g = torch.Generator().manual_seed(M + N + K + act)
    x = (torch.randn(M, K, generator=g) * (0.5 + 2.0 * torch.rand(M, 1, generator=g)) + 1.5 * torch.randn(M, 1, generator=g)).bfloat16()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16()
    gamma, beta, bias = 1.0 + 0.2 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g), 0.1 * torch.randn(N, generator=g)
    xd, wd = x.to(DEV), w.to(DEV)
    stats = ops.row_stats(xd, 1e-5)
    mean = x.double().mean(1); var = x.double().var(1, unbiased=False)
    assert _rel_err(stats[:, 5].cpu(), mean) < 1e-5 and _rel_err(stats[:, 4].cpu(), (var + 1e-5).rsqrt()) < 1e-5
    folded = ops.ln_fold(wd, gamma.to(DEV), beta.to(DEV), bias.to(DEV))
    assert torch.equal(folded[0].cpu(), (w.float() * gamma).bfloat16())
    assert _rel_err(folded[1].cpu(), (w.float() * gamma).bfloat16().double().sum(1)) < 1e-5
    assert _rel_err(folded[2].cpu(), bias.double() + w.double() @ beta.double()) < 1e-5
    got = ops.linear_ln(xd, folded, stats, act=act).float().cpu()
    unfolded = ops.linear(ops.layernorm(xd, gamma.to(DEV), beta.to(DEV), 1e-5), wd, bias.to(DEV), act=act).float().cpu()
    y = (x.double() - mean[:, None]) * (var[:, None] + 1e-5).rsqrt() * gamma.double() + beta.double()
    ref = y @ w.double().t() + bias.double()
    if act == 1:
        ref = ref * torch.sigmoid(1.702 * ref)
    elif act == 2:
        ref = F.gelu(ref)
    e_f, e_u = _rel_err(got, ref), _rel_err(unfolded, ref)
    assert e_f < 8e-3 and e_f <= 1.5 * e_u + 1e-3, (e_f, e_u)


def test_linear_ln_rows_do_not_depend_on_the_kernel():
    """The persistent 256 x 256 kernel (a batch of images) and the 64 x 64 kernel (a few images) must give a row the same bits: same fragments
    for the rank-2 start of the accumulators, same k order, same epilogue."""
    g = torch.Generator().manual_seed(77)
    M, N, K = 65792 // 4, 3072, 1024
    x = (torch.randn(M, K, generator=g) + 0.7).bfloat16().to(DEV)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().to(DEV)
    gamma, beta, bias = (1.0 + 0.1 * torch.randn(K, generator=g)).to(DEV), (0.1 * torch.randn(K, generator=g)).to(DEV), (0.1 * torch.randn(N, generator=g)).to(DEV)
    folded = ops.ln_fold(w, gamma, beta, bias)
    stats = ops.row_stats(x, 1e-5)
    for act in (0, 1):
        big = ops.linear_ln(x, folded, stats, act=act)                       # 65 x 12 tiles: persistent (+ its 64 x 64 remainder launch)
        for lo, n in ((0, 257), (5000, 771), (M - 300, 300)):
            small = ops.linear_ln(x[lo:lo + n].contiguous(), folded, stats[lo:lo + n].contiguous(), act=act)
            assert torch.equal(small, big[lo:lo + n]), (act, lo)


# ---- 256 < N <= 576 (cfg4: 24 x 24 patches): the strip kernel — one launch, workgroups of one image exchanging rho / row max / scores ------------
STRIP_CASES = [
    # N, C, B, k, mcn, thr, noise, mask
    (576, 1024, 3, 64, 64, 0.5, False, False),
    (576, 1024, 3, 64, 64, 1e9, True, False),          # the top-min_cluster_num fallback
    (576, 1024, 2, 8, 64, 0.3, False, True),           # token_mask: the global raw maximum is a third exchange
    (576, 256, 21, 2, 5, 0.5, True, False),            # more images than one round of 8 queues x a few workgroups; k = 2
    (400, 512, 4, 64, 32, 0.4, False, False),          # 20 x 20: columns and rows beyond N, a partial column tile (400 = 25 x 16)
    (324, 768, 3, 324, 16, 0.5, False, True),          # 18 x 18: 324 = 20.25 column tiles, k = N, three strips (the last one 68 rows)
    (257, 64, 5, 16, 300 - 257 + 1, 0.2, True, False), # the smallest N of this path
]


@pytest.mark.parametrize("N,C,B,k,mcn,thr,with_noise,with_mask", STRIP_CASES)
def test_cluster_strips_equal_the_multi_kernel_form_and_the_oracle(N, C, B, k, mcn, thr, with_noise, with_mask):
    g = torch.Generator().manual_seed(N * 7 + C + B)
    m = 6
    xs = []
    for i in range(B):
        gx = torch.Generator().manual_seed(700 + 13 * i + N)
        cent = torch.randn(m + i % 5, C, generator=gx) * 2.0
        xs.append(cent[torch.randint(0, cent.shape[0], (N,), generator=gx)] + 0.05 * torch.randn(N, C, generator=gx))
    xs = torch.stack(xs).bfloat16()
    noise = torch.rand(B, N, generator=g) if with_noise else None
    mask = None
    if with_mask:
        mask = (torch.rand(B, N, generator=g) > 0.3).float()
        mask[:, 0] = 1.0
    assert os.environ.get("SETOK_CLUSTER_FUSED") is None
    a = ops.cluster_dpc_knn(xs.to(DEV).reshape(-1, C), B, N, k, thr, mcn, noise, mask)
    a2 = ops.cluster_dpc_knn(xs.to(DEV).reshape(-1, C), B, N, k, thr, mcn, noise, mask)
    assert all(torch.equal(u, v) for u, v in zip(a, a2))                                     # run to run: identical bits
    os.environ["SETOK_CLUSTER_FUSED"] = "0"
    try:
        bm = ops.cluster_dpc_knn(xs.to(DEV).reshape(-1, C), B, N, k, thr, mcn, noise, mask)
    finally:
        del os.environ["SETOK_CLUSTER_FUSED"]
    idx, score, index_down, counts = a
    wide = k == N
    for i in range(B):
        L = int(counts[i])
        nz = None if noise is None else noise[i]
        tm = None if mask is None else mask[i]
        r = O.cluster_dpc_knn(xs[i].float(), k, thr, mcn, tm, nz)
        sens = O.cluster_sensitivity(xs[i].float(), k, thr, mcn, tm, nz, ulps=4096.0 if wide else 64.0)
        assert 1 <= L <= N and int(idx[i].max()) < L and int(idx[i].min()) >= 0 and bool((index_down[i, L:] == -1).all()) and bool(torch.isfinite(score[i]).all())
        O.check_cluster_parity(index_down[i, :L].cpu(), idx[i].cpu(), r.index_down, r.idx_cluster, sens)
        if not wide:
            O.check_score(score[i].cpu(), sens)
        if sens["centres_certain"]:                                                          # the two GPU forms select alike wherever the decision is certain
            assert L == int(bm[3][i]) and torch.equal(index_down[i, :L], bm[2][i, :L])
    one = ops.cluster_dpc_knn(xs[B - 1].to(DEV), 1, N, k, thr, mcn, None if noise is None else noise[B - 1:], None if mask is None else mask[B - 1:])
    assert torch.equal(one[0][0], idx[B - 1]) and torch.equal(one[1][0], score[B - 1]) and int(one[3][0]) == int(counts[B - 1])    # image alone == in the batch


def test_cluster_strips_full_batch_cfg4():
    """cfg4's clustering call at full size (128 images x 576 tokens x 1024 channels = 640 (image, strip) items on a persistent grid of one
    workgroup per CU): every image equal to the same image clustered alone (the exchanges never mix images up), twice the same bits."""
    B, N, C = 128, 576, 1024
    g = torch.Generator().manual_seed(1)
    cent = torch.randn(40, C, generator=g) * 1.5
    xs = (cent[torch.randint(0, 40, (B, N), generator=g)] + 0.2 * torch.randn(B, N, C, generator=g)).bfloat16().to(DEV)
    a = ops.cluster_dpc_knn(xs.reshape(-1, C), B, N, 64, 0.125, 64)
    b = ops.cluster_dpc_knn(xs.reshape(-1, C), B, N, 64, 0.125, 64)
    assert all(torch.equal(u, v) for u, v in zip(a, b))
    for i in (0, 1, 7, 8, 63, 126, 127):
        one = ops.cluster_dpc_knn(xs[i], 1, N, 64, 0.125, 64)
        assert torch.equal(one[0][0], a[0][i]) and torch.equal(one[1][0], a[1][i]) and torch.equal(one[2][0], a[2][i]) and int(one[3][0]) == int(a[3][i]), i
    assert int(a[3].min()) >= 1 and int(a[0].max()) < int(a[3].max())


@pytest.mark.parametrize("cap", [32, 33, 9, 5])
def test_cluster_strips_on_a_small_grid_complete_and_keep_their_bits(cap):
    """ADVICE r03 / VERDICT r03 item 6: the strips of an image spin-wait on each other, so the number of item queues must follow the number of
    co-resident workgroups (grid > queues x (strips - 1)).  SETOK_STRIP_GRID caps the persistent grid the way a 32-CU partition or a CU mask would:
    32 workgroups at 5 strips used to leave 4 home workgroups per queue waiting for a fifth strip nobody could pull.  The results must not depend
    on the grid (every item's arithmetic is its own; the exchanges carry the same numbers)."""
    B, N, C = 24, 576, 256
    g = torch.Generator().manual_seed(5)
    cent = torch.randn(30, C, generator=g) * 1.5
    xs = (cent[torch.randint(0, 30, (B, N), generator=g)] + 0.2 * torch.randn(B, N, C, generator=g)).bfloat16().to(DEV)
    full = ops.cluster_dpc_knn(xs.reshape(-1, C), B, N, 64, 0.125, 64)
    torch.cuda.synchronize()
    assert os.environ.get("SETOK_STRIP_GRID") is None
    os.environ["SETOK_STRIP_GRID"] = str(cap)
    try:
        small = ops.cluster_dpc_knn(xs.reshape(-1, C), B, N, 64, 0.125, 64)
        torch.cuda.synchronize()
    finally:
        del os.environ["SETOK_STRIP_GRID"]
    assert all(torch.equal(u, v) for u, v in zip(full, small))


def test_cluster_strips_refuse_a_grid_that_cannot_hold_one_image():
    B, N, C = 2, 576, 64
    xs = torch.randn(B * N, C).bfloat16().to(DEV)
    os.environ["SETOK_STRIP_GRID"] = "4"                                   # 5 strips per image
    try:
        with pytest.raises(RuntimeError, match="usable workgroups"):
            ops.cluster_dpc_knn(xs, B, N, 64, 0.125, 64)
    finally:
        del os.environ["SETOK_STRIP_GRID"]


def _with_env(name, value, fn):
    assert os.environ.get(name) is None
    os.environ[name] = value
    try:
        return fn()
    finally:
        del os.environ[name]


@pytest.mark.parametrize("T,B", [(257, 1), (257, 3), (577, 1), (64, 2), (197, 2), (129, 1), (256, 2), (288, 1), (145, 5)])
def test_vit_attention_query_split_keeps_its_bits(T, B):
    """Round 4, small batches: a handful of (image, head) pairs cannot occupy the chip, so the query tiles of a pair are split over several
    workgroups (each stages the head's K / V).  A tile's arithmetic does not depend on who runs it: every split gives the same bits — for the
    online-softmax kernel and for the opt-in row-resident kernel of 129 <= T <= 288 (SETOK_ATTN_ROW=1: 16-query tiles, the exact row maximum;
    at other lengths the switch changes nothing)."""
    H, Dh = 16, 64
    g = torch.Generator().manual_seed(T + B)
    qkv = torch.randn(B * T, 3 * H * Dh, generator=g).bfloat16().to(DEV)
    run = lambda: ops.attention(qkv, H, Dh, Dh ** -0.5, seg_len=T)
    assert os.environ.get("SETOK_ATTN_QSPLIT") is None and os.environ.get("SETOK_ATTN_ROW") is None
    q, k, v = (t.reshape(B, T, H, Dh).transpose(1, 2) for t in qkv.float().cpu().split(H * Dh, dim=1))
    ref = (torch.softmax(q @ k.transpose(-1, -2) * Dh ** -0.5, dim=-1) @ v).transpose(1, 2).reshape(B * T, H * Dh)
    for row in (None, "1"):
        form = (lambda f: f()) if row is None else (lambda f: _with_env("SETOK_ATTN_ROW", row, f))
        auto = form(run)
        for qs in (1, 2, 5, 64):
            assert torch.equal(form(lambda: _with_env("SETOK_ATTN_QSPLIT", str(qs), run)), auto), (row, qs)
        assert _rel_err(auto.float().cpu(), ref) < 2e-2                    # and they are the right bits
        if row is None:                                                    # the head-pair experiment (a workgroup per pair of adjacent heads of the online kernel): same bits
            assert torch.equal(form(lambda: _with_env("SETOK_ATTN_HEADPAIR", "1", run)), auto)


def test_vit_attention_row_kernel_is_batch_invariant_and_closer_to_fp32():
    """The opt-in row-resident kernel (SETOK_ATTN_ROW=1) keeps a query tile's whole score row in registers: the softmax uses the exact row maximum
    (no running rescale), an image's rows do not depend on the batch it is in, and it is no further from the fp32 result than the default kernel."""
    H, Dh, T = 16, 64, 257
    g = torch.Generator().manual_seed(5)
    qkv = (torch.randn(6 * T, 3 * H * Dh, generator=g) * 2.0).bfloat16().to(DEV)
    row = lambda f: _with_env("SETOK_ATTN_ROW", "1", f)
    full = row(lambda: ops.attention(qkv, H, Dh, Dh ** -0.5, seg_len=T))
    for i in (0, 3, 5):
        one = row(lambda: ops.attention(qkv[i * T:(i + 1) * T].contiguous(), H, Dh, Dh ** -0.5, seg_len=T))
        assert torch.equal(one, full[i * T:(i + 1) * T])
    q, k, v = (t.reshape(6, T, H, Dh).transpose(1, 2) for t in qkv.float().cpu().split(H * Dh, dim=1))
    ref = (torch.softmax(q @ k.transpose(-1, -2) * Dh ** -0.5, dim=-1) @ v).transpose(1, 2).reshape(6 * T, H * Dh)
    old = ops.attention(qkv, H, Dh, Dh ** -0.5, seg_len=T)
    assert not torch.equal(old, full)                                      # (another kernel: other bits)
    e_new, e_old = _rel_err(full.float().cpu(), ref), _rel_err(old.float().cpu(), ref)
    assert e_new < 1e-2 and e_new <= e_old * 1.05, (e_new, e_old)


def test_cluster_strips_on_a_cu_masked_stream():
    """ADVICE r03, the real thing: a stream created with a 32-CU mask (hipExtStreamCreateWithCUMask) admits 32 co-resident workgroups of the
    persistent strip kernel, whatever hipDeviceAttributeMultiprocessorCount says.  The launch must read the mask (hipExtStreamGetCUMask), size its grid
    and its queue count by it, complete, and give the bits of the unmasked launch."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    B, N, Cc = 24, 576, 256
    g = torch.Generator().manual_seed(6)
    cent = torch.randn(30, Cc, generator=g) * 1.5
    xs = (cent[torch.randint(0, 30, (B, N), generator=g)] + 0.2 * torch.randn(B, N, Cc, generator=g)).bfloat16().to(DEV)
    full = ops.cluster_dpc_knn(xs.reshape(-1, Cc), B, N, 64, 0.125, 64)
    torch.cuda.synchronize()
    mask = (C.c_uint32 * 8)(0xFFFFFFFF, 0, 0, 0, 0, 0, 0, 0)                 # 32 of the 256 CUs
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, mask)
    if rc != 0:
        pytest.skip(f"hipExtStreamCreateWithCUMask failed ({rc})")
    try:
        ext = torch.cuda.ExternalStream(st.value)
        with torch.cuda.stream(ext):
            small = ops.cluster_dpc_knn(xs.reshape(-1, Cc), B, N, 64, 0.125, 64)
        ext.synchronize()
    finally:
        hip.hipStreamDestroy(st)
    assert all(torch.equal(u, v) for u, v in zip(full, small))


_MERGE_REM_CASES = [(act, res, ln) for act in (0, 1, 2) for res in (False, True) for ln in (False, True) if not (res and ln)]


def _merge_rem_outputs():
    """Every activation x residual x folded-LayerNorm combination of the ping-pong GEMM at M = 257 tile rows + 3 rows (the ViT's shape class: whole 256-row
    tiles + a remainder), N = 1024 (32 x 32 remainder tiles) and N = 3072 (64 x 64): outputs as a list of CPU tensors."""
    outs = []
    for N, K in ((1024, 1024), (3072, 768)):
        M = 256 * 24 + 259
        a = (_rand(M, K, seed=21) * 1.3 + 0.2).bfloat16().to(DEV)
        w = _rand(N, K, seed=22, scale=K ** -0.5).bfloat16().to(DEV)
        b = _rand(N, seed=23).to(DEV)
        r = _rand(M, N, seed=24).bfloat16().to(DEV)
        gamma, beta = (1.0 + 0.1 * _rand(K, seed=25)).to(DEV), (0.1 * _rand(K, seed=26)).to(DEV)
        for act, res, ln in _MERGE_REM_CASES:
            if ln:
                outs.append(ops.linear_ln(a, ops.ln_fold(w, gamma, beta, b), ops.row_stats(a, 1e-5), act=act).cpu())
            else:
                outs.append(ops.linear(a, w, b, r if res else None, act=act).cpu())
    return outs


def test_remainder_rows_inside_the_launch_equal_the_separate_launch(tmp_path):
    """since round 5 a ping-pong launch finishes the rows behind its last whole 256-row tile itself (PP_MERGE_REM; the first wave row runs the
    small-tile kernel's tile function while the second has ended — correct only because s_barrier counts the surviving waves).  SETOK_GEMM_MERGE_REM=0 sends
    them to a launch of the small-tile kernel as rounds 3-4 did.  The switch is read once per process, so the other arm runs in a child process: identical
    bits for every activation x residual x folded-LayerNorm combination, at both remainder tile shapes."""
    import subprocess, sys
    here = _merge_rem_outputs()
    path = str(tmp_path / "separate.pt")
    code = ("import sys, torch; sys.path[:0] = [%r, %r]; import test_ops_gpu as t; torch.set_grad_enabled(False); torch.save(t._merge_rem_outputs(), %r)"
            % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))), path))
    env = dict(os.environ, SETOK_GEMM_MERGE_REM="0", PYTHONPATH=os.pathsep.join(sys.path))
    subprocess.run([sys.executable, "-c", code], check=True, env=env, timeout=600)
    there = torch.load(path)
    assert len(here) == len(there) == 2 * len(_MERGE_REM_CASES)
    for i, (x, y) in enumerate(zip(here, there)):
        assert torch.equal(x, y), (i, _MERGE_REM_CASES[i % len(_MERGE_REM_CASES)])
