"""§8(f) row 4 on a real MI355X: the hand-written backward pass of the trainable head (csrc/backward.hip + setok_linear) and the
AdamW step, through the C ABI, against (i) the gradients of the REFERENCE's modules under the reference's own autograd
(tests/golden/head_grads.npz) and (ii) torch-autograd fp64 references of each backward op.  `pytest -m gpu`."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import setok_oracle as O

pytestmark = [pytest.mark.gpu, pytest.mark.grad]        # torch-autograd references + the hand-written backward: gradients stay enabled

if torch.cuda.is_available():
    import setok_amd
    from setok_amd import ops, SetokTokenizer
    from setok_amd.training import HeadTrainer, head_backward, head_forward_train, SITE_STRIDE

DEV = "cuda"


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _rel(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


# ---- the backward ops one by one -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,cols,pad", [(5, 8, 1), (130, 96, 16), (1000, 1024, 64), (64, 72, 64)])
def test_transpose_pads_with_zeros(dt, rows, cols, pad):
    x = _rand(rows, cols, seed=1).to(dt)
    out = ops.transpose(x.to(DEV), pad).cpu()
    ldo = (rows + pad - 1) // pad * pad
    assert out.shape == (cols, ldo) and torch.equal(out[:, :rows], x.t()) and float(out[:, rows:].abs().max() if ldo > rows else 0) == 0.0
    out2, cs = ops.transpose(x.to(DEV), pad, with_colsum=True)
    assert torch.equal(out2.cpu(), out) and _rel(cs, x.double().sum(0)) < 1e-5


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-5), (torch.float16, 2e-5)])
@pytest.mark.parametrize("M,N,K,S", [(5000, 96, 64, 4), (20000, 256, 128, 8), (700, 64, 64, 3)])
def test_split_k_weight_gradient(dt, tol, M, N, K, S):
    """dW = dY^T X through the split-K layout (chunked transposes, one batched GEMM, fixed-order sum) == the plain product."""
    dy, x = _rand(M, N, seed=20).to(dt), _rand(M, K, seed=21).to(dt)
    pad = 16 if dt == torch.float32 else 64
    aT, bT = ops.transpose(dy.to(DEV), pad, S), ops.transpose(x.to(DEV), pad, S)
    assert aT.shape[0] == S and aT.shape[1] == N and aT.shape[2] % pad == 0
    got = ops.linear_tn(aT, bT)
    assert got.dtype == torch.float32 and _rel(got, dy.double().t() @ x.double()) < tol
    again = ops.linear_tn(aT, bT)
    assert torch.equal(got, again)


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-5), (torch.float16, 1e-5)])
def test_colsum_and_accumulate(dt, tol):
    x = _rand(3000, 200, seed=2).to(dt)
    out = ops.colsum(x.to(DEV))
    assert _rel(out, x.double().sum(0)) < tol
    ops.colsum(x.to(DEV), out=out, accumulate=True)
    assert _rel(out, 2 * x.double().sum(0)) < tol


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2), (torch.float16, 2e-2)])
@pytest.mark.parametrize("rows,C", [(7, 64), (1030, 1024), (33, 768)])
def test_layernorm_bwd(dt, tol, rows, C):
    x, dy, res = (_rand(rows, C, seed=3) * 2 + 0.3).to(dt), _rand(rows, C, seed=4).to(dt), _rand(rows, C, seed=5).to(dt)
    gam = 1 + 0.1 * _rand(C, seed=6)
    xr = x.double().requires_grad_(True); gr = gam.double().requires_grad_(True); br = torch.zeros(C, dtype=torch.float64, requires_grad=True)
    F.layer_norm(xr, (C,), gr, br, 1e-5).backward(dy.double())
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dx = ops.layernorm_bwd(x.to(DEV), dy.to(DEV), gam.to(DEV), 1e-5, dg, db, accumulate=False, res=res.to(DEV))
    assert _rel(dx, xr.grad + res.double()) < tol
    assert _rel(dg, gr.grad) < max(tol, 1e-4) and _rel(db, br.grad) < max(tol, 1e-4)
    ops.layernorm_bwd(x.to(DEV), dy.to(DEV), gam.to(DEV), 1e-5, dg, db, accumulate=True, need_dx=False)
    assert _rel(dg, 2 * gr.grad) < max(tol, 1e-4)


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2), (torch.float16, 1e-2)])
def test_gelu_bwd(dt, tol):
    pre, dy = (_rand(4000, seed=7) * 2).to(dt), _rand(4000, seed=8).to(dt)
    pr = pre.double().requires_grad_(True)
    F.gelu(pr).backward(dy.double())
    assert _rel(ops.gelu_bwd(pre.to(DEV), dy.to(DEV)), pr.grad) < tol


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-5), (torch.bfloat16, 3e-2), (torch.float16, 3e-2)])
@pytest.mark.parametrize("H,Dh,lens", [(2, 32, [1, 7, 64, 3, 20]), (2, 512, [5, 1, 9]), (4, 16, [130, 2]), (2, 512, [40, 130, 7, 33, 64, 32, 229])])
def test_attention_bwd_ragged(dt, tol, H, Dh, lens):
    offs = np.concatenate([[0], np.cumsum(lens)]).tolist()
    C = H * Dh
    qkv = _rand(offs[-1], 3 * C, seed=9).to(dt)
    dout = _rand(offs[-1], C, seed=10).to(dt)
    scale = Dh ** -0.5
    qr = qkv.double().requires_grad_(True)
    outs = []
    for s in range(len(lens)):
        blk = qr[offs[s]:offs[s + 1]].reshape(-1, 3, H, Dh).permute(1, 2, 0, 3)
        a = torch.softmax(blk[0] @ blk[1].transpose(-1, -2) * scale, -1)
        outs.append((a @ blk[2]).transpose(0, 1).reshape(-1, C))
    out = torch.cat(outs, 0)
    out.backward(dout.double())
    so = torch.tensor(offs, dtype=torch.int32, device=DEV)
    o_dev = ops.attention(qkv.to(DEV), H, Dh, scale, seg_len=max(lens), seg_offsets=so, n_segs=len(lens))
    dqkv = ops.attention_bwd(qkv.to(DEV), o_dev, dout.to(DEV), H, Dh, scale, max(lens), so, len(lens))
    assert _rel(dqkv, qr.grad) < tol


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
def test_segment_mean_bwd(dt):
    lens = [3, 1, 6, 2]
    offs = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device=DEV)
    n = torch.tensor([len(lens)], dtype=torch.int32, device=DEV)
    dseg = _rand(len(lens), 64, seed=11).to(dt)
    got = ops.segment_mean_bwd(dseg.to(DEV), offs, n, len(lens), sum(lens)).cpu()
    ref = torch.cat([(dseg[i].float() / l).to(dt).expand(l, 64) for i, l in enumerate(lens)], 0)
    assert torch.equal(got, ref)


def test_adamw_matches_torch():
    p0, g = _rand(1000, seed=12), _rand(1000, seed=13)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
    p, m, v = p0.clone().to(DEV), torch.zeros(1000, device=DEV), torch.zeros(1000, device=DEV)
    lp = torch.empty(1000, device=DEV, dtype=torch.bfloat16)
    for step in (1, 2, 3):
        ref.grad = g * step
        opt.step()
        ops.adamw(p, (g * step * 2).to(DEV), m, v, lp, 1e-2, 0.9, 0.95, 1e-8, 0.1, step, grad_scale=0.5)
    assert _rel(p, ref.data) < 1e-6 and torch.equal(lp.cpu(), p.cpu().bfloat16())


# ---- the head: gradients vs the reference's own autograd ------------------------------------------------------------------
def _small_tok(sd, dtype=torch.float32):
    vc = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=4, image_size=112, patch_size=14)
    tok = SetokTokenizer(vision_tower=vc, mm_vision_select_layer=-2, hidden_dim=64, token_feat_dim=96, min_cluster_num=8,
                         threshold=0.5, nheads=2, dim_feedforward=128)
    assert not tok.load_state_dict(sd, strict=False).unexpected_keys
    return tok.to(device=DEV, dtype=dtype).eval()


def _grad_case(golden_dir):
    z = np.load(os.path.join(golden_dir, "head_small.npz"))
    gz = np.load(os.path.join(golden_dir, "head_grads.npz"))
    sd = {k[2:]: _t(z[k]) for k in z.files if k.startswith("w:")}
    feats = [_t(z["dynamic:feats"]), _t(z["planted:feats"])]
    hidden = torch.cat([torch.cat([torch.zeros(1, 64), f], 0) for f in feats], 0)          # class-token rows that 'patch' drops
    ups = [_t(gz["up:0"]), _t(gz["up:1"])]
    ref = {k[2:]: _t(gz[k]) for k in gz.files if k.startswith("g:")}
    return sd, hidden, ups, ref, float(gz["threshold"]), gz["counts"].tolist()


def test_head_gradients_match_reference_autograd(golden_dir):
    sd, hidden, ups, ref, thr, counts = _grad_case(golden_dir)
    tok = _small_tok(sd)
    tokens, ctx = head_forward_train(tok, hidden.to(DEV), 2, threshold=thr)
    assert tokens.counts == counts
    inf, _, _ = tok.encode_features(hidden.to(DEV), 2, threshold=thr)                     # the training forward IS the inference forward (fp32)
    assert _rel(tokens.packed, inf.packed) < 1e-6
    seen = []
    grads = head_backward(tok, ctx, torch.cat(ups, 0).to(DEV), on_module_done=lambda m, g: seen.append((m, sorted(g))))
    assert [m for m, _ in seen] == ["out", "inter_encoder", "inner_encoder"] and all(len(n) > 0 for _, n in seen)
    assert set(grads) == set(ref)
    worst = max(_rel(grads[n], ref[n]) for n in ref)
    assert worst < 1e-4, {n: _rel(grads[n], ref[n]) for n in ref if _rel(grads[n], ref[n]) >= 1e-4}
    again = head_backward(tok, ctx, torch.cat(ups, 0).to(DEV))                            # deterministic: no atomics anywhere
    assert all(torch.equal(grads[n], again[n]) for n in grads)


def test_trainer_step_matches_torch_adamw(golden_dir):
    """One full step (forward, backward, AdamW) in fp32 against torch.optim.AdamW applied to the same gradients (the gradients
    themselves are checked against the reference above; the key-bias gradient is exactly 0 in exact arithmetic — softmax is
    shift-invariant — so its rounding-noise sign, which is all a first Adam step sees, is not comparable across implementations)."""
    sd, hidden, ups, ref, thr, _ = _grad_case(golden_dir)
    tok = _small_tok(sd)
    tr = HeadTrainer(tok, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, dropout="eval")
    before = {n: p.detach().clone() for n, p in tr.params.items()}
    _, ctx = head_forward_train(tok, hidden.to(DEV), 2, threshold=thr)
    tr.backward(ctx, torch.cat(ups, 0).to(DEV))
    tr.step()
    for n, p in tr.params.items():
        q = torch.nn.Parameter(before[n].cpu().clone()); q.grad = tr.grads[n].cpu().reshape(q.shape).clone()
        torch.optim.AdamW([q], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01).step()
        assert _rel(p.detach(), q.data) < 1e-5, n
    # the updated weights are what the next forward uses
    t2, _ = head_forward_train(tok, hidden.to(DEV), 2, threshold=thr)
    ref_sd = {k: v.clone() for k, v in sd.items()}
    for n, p in tr.params.items():
        ref_sd[n] = p.detach().cpu()
    hc = O.HeadConfig(hidden_dim=64, token_feat_dim=96, min_cluster_num=8, threshold=0.5, nheads=2, dim_feedforward=128)
    want = O.head_forward(ref_sd, hc, hidden[1:65], None, thr).tokens
    assert _rel(t2[0], want) < 1e-4


# ---- training-mode dropout (module.py:36,44,45,59,72) ------------------------------------------------------------------------
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
def test_dropout_op_is_a_seeded_bernoulli_mask(dt):
    """setok_dropout: keep-rate 1 - p (within 5 sigma), survivors scaled by 1 / (1 - p), a pure function of (seed, offset + i) — the same call on
    a gradient is the backward pass —, residual form, in-place form."""
    n, p = 1 << 20, 0.2
    x = (_rand(n, seed=3) + 3.0).to(dt).to(DEV)
    y = ops.dropout(x, p, seed=77, offset=5)
    kept = y != 0
    rate = float(kept.float().mean())
    assert abs(rate - (1 - p)) < 5 * (p * (1 - p) / n) ** 0.5
    assert torch.equal(y[kept], (x.float()[kept] * (1.0 / (1.0 - p))).to(dt))
    assert torch.equal(ops.dropout(x, p, seed=77, offset=5), y)                                  # deterministic
    assert not torch.equal(ops.dropout(x, p, seed=78, offset=5) != 0, kept)                      # another seed, another mask
    m0 = ops.dropout(torch.ones(n, device=DEV), p, seed=77, offset=0) != 0
    assert torch.equal(m0[5:], kept[:-5])                                                        # the counter is offset + i
    r = _rand(n, seed=4).to(dt).to(DEV)
    z = ops.dropout(x, p, seed=77, offset=5, residual=r)
    assert torch.equal(z, (r.float() + torch.where(kept, x.float() * (1.0 / (1.0 - p)), torch.zeros_like(x, dtype=torch.float32))).to(dt))   # one rounding
    x2 = x.clone()
    assert ops.dropout(x2, p, seed=77, offset=5, out=x2) is x2 and torch.equal(x2, y)
    assert torch.equal(ops.dropout(x, 0.0, seed=1), x)
    xo = x[3:]                                                                                   # a view that is not 16-byte aligned: the element-wise kernel, the same mask
    assert torch.equal(ops.dropout(xo, p, seed=77, offset=8), y[3:])
    from setok_amd._lib import SetokHipError
    with pytest.raises(SetokHipError):
        ops.dropout(x, 1.0, seed=1)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
def test_fused_dropout_passes_equal_the_two_launch_forms(dt):
    """Round 4: drop(act(x)) and gelu'(pre) * drop(dy) as ONE pass each (the Mlp's middle dropout, module.py:41,44): the intermediate is rounded to
    the storage type exactly as the two-launch forms round it — identical bits, at aligned and unaligned counter offsets, with a ragged tail."""
    for n, off in ((1 << 18, 0), ((1 << 16) + 5, 7), (13, 1 << 40)):
        x = (_rand(n, seed=5) * 2.0).to(dt).to(DEV)
        g = _rand(n, seed=6).to(dt).to(DEV)
        ref = ops.dropout(ops.activation(x, ops.ACT_GELU_ERF), 0.2, seed=9, offset=off)
        assert torch.equal(ops.activation_dropout(x, ops.ACT_GELU_ERF, 0.2, seed=9, offset=off), ref)
        ref_b = ops.gelu_bwd(x, ops.dropout(g, 0.2, seed=9, offset=off))
        assert torch.equal(ops.gelu_bwd_dropout(x, g, 0.2, seed=9, offset=off), ref_b)
        g2 = g.clone()
        assert ops.gelu_bwd_dropout(x, g2, 0.2, seed=9, offset=off, out=g2) is g2 and torch.equal(g2, ref_b)       # in place over the gradient


def _block_with_masks(sd, prefix, x, bounds, nheads, depth, masks):
    """Block.forward in TRAINING mode on packed rows (segments = `bounds`), torch autograd, the dropout masks given: masks[i] multiplies the
    attention projection of layer i, masks[depth] the Mlp activation, masks[depth + 1] the fc2 output (module.py:40-45,71-72,96-98)."""
    n, C = x.shape
    dh = C // nheads
    g1, b1 = sd[prefix + "norm1.weight"], sd[prefix + "norm1.bias"]
    for i in range(depth):
        a = prefix + f"layers.{i}.1."
        y = F.layer_norm(x, (C,), g1, b1, 1e-5)
        qkv = F.linear(y, sd[a + "qkv.weight"], sd[a + "qkv.bias"])
        outs = []
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            q, kk, v = qkv[lo:hi].reshape(hi - lo, 3, nheads, dh).permute(1, 2, 0, 3)
            att = torch.softmax((q @ kk.transpose(-2, -1)) * (dh ** -0.5), dim=-1)
            outs.append((att @ v).transpose(0, 1).reshape(hi - lo, C))
        x = x + masks[i] * F.linear(torch.cat(outs, 0), sd[a + "proj.weight"], sd[a + "proj.bias"])
    y = F.layer_norm(x, (C,), sd[prefix + "norm2.weight"], sd[prefix + "norm2.bias"], 1e-5)
    u = masks[depth] * F.gelu(F.linear(y, sd[prefix + "mlp.fc1.weight"], sd[prefix + "mlp.fc1.bias"]))
    return x + masks[depth + 1] * F.linear(u, sd[prefix + "mlp.fc2.weight"], sd[prefix + "mlp.fc2.bias"])


def test_head_gradients_with_dropout_match_autograd(golden_dir):
    """The training-mode step of the head (masks active at the three proj_drop sites of both Blocks) against torch autograd (fp64) through the same
    arithmetic with the SAME masks (read back from setok_dropout): tokens and all 34 parameter gradients."""
    sd, hidden, ups, _, thr, counts = _grad_case(golden_dir)
    tok = _small_tok(sd)
    p, seed = 0.2, 12345
    tok.inner_encoder.proj_drop_p = tok.inter_encoder.proj_drop_p = p
    tokens, ctx = head_forward_train(tok, hidden.to(DEV), 2, threshold=thr, dropout_seed=seed)
    assert tokens.counts == counts
    ev, _ = head_forward_train(tok, hidden.to(DEV), 2, threshold=thr)
    assert _rel(tokens.packed, ev.packed) > 1e-2                                                 # the masks did something
    grads = head_backward(tok, ctx, torch.cat(ups, 0).to(DEV))
    # ---- the reference: same packed rows, same segments, same masks
    hs = ctx["inner"]["x"][0].double().cpu()
    seg = ctx["seg_offsets"].cpu().tolist()[: ctx["total"] + 1]
    img = ctx["img_offsets"].cpu().tolist()
    d_in, d_it = len(tok.inner_encoder._pack()["attn"]), len(tok.inter_encoder._pack()["attn"])
    C, FF, rows, L = 64, 128, hs.shape[0], ctx["total"]

    def factors(site0, depth, n):
        widths = [C] * depth + [FF, C]
        return [ops.dropout(torch.ones(n, w, device=DEV), p, seed, (site0 + i) * SITE_STRIDE).double().cpu() for i, w in enumerate(widths)]
    m_in, m_it = factors(0, d_in, rows), factors(d_in + 2, d_it, L)
    params = {n: v.double().clone().requires_grad_(True) for n, v in sd.items() if n.split(".")[0] in ("inner_encoder", "inter_encoder", "out")
              and not (".layers." in n and n.split(".layers.")[1].split(".")[1] == "0")}
    for n in [k for k in sd if ".layers." in k and k.split(".layers.")[1].split(".")[1] == "0"]:        # norm1 aliases (module.py:87-88)
        params[n] = params[n.split(".layers.")[0] + ".norm1." + n.split(".")[-1]]
    inner = _block_with_masks(params, "inner_encoder.", hs, seg, 2, d_in, m_in)
    group = torch.stack([inner[lo:hi].mean(0) for lo, hi in zip(seg[:-1], seg[1:])], 0)
    inter = _block_with_masks(params, "inter_encoder.", group, img[: 3], 2, d_it, m_it)
    want = F.linear(inter, params["out.weight"], params["out.bias"])
    assert _rel(tokens.packed, want) < 1e-5
    (want * torch.cat(ups, 0).double()).sum().backward()
    ref = {n: q.grad for n, q in params.items() if q.grad is not None and not (".layers." in n and n.split(".layers.")[1].split(".")[1] == "0")}
    assert set(grads) == set(ref)
    worst = {n: _rel(grads[n], ref[n]) for n in ref}
    assert max(worst.values()) < 1e-4, {n: e for n, e in worst.items() if e >= 1e-4}
    # same seed, same step; another seed, another step
    t2, _ = head_forward_train(tok, hidden.to(DEV), 2, threshold=thr, dropout_seed=seed)
    t3, _ = head_forward_train(tok, hidden.to(DEV), 2, threshold=thr, dropout_seed=seed + 1)
    assert torch.equal(t2.packed, tokens.packed) and not torch.equal(t3.packed, tokens.packed)


def test_trainer_dropout_policies(golden_dir):
    """HeadTrainer(dropout=...): "train" draws new masks every step from (seed, step, rank) and repeats them for the same (seed, step); attn_drop > 0
    raises; "error" refuses a module built with proj_drop > 0; "eval" is the unregularised forward."""
    sd, hidden, ups, _, thr, _ = _grad_case(golden_dir)
    tok = _small_tok(sd)
    tok.inner_encoder.proj_drop_p = tok.inter_encoder.proj_drop_p = 0.2
    with pytest.raises(NotImplementedError):
        HeadTrainer(tok, dropout="error")
    tr = HeadTrainer(tok, lr=1e-3, dropout="train", dropout_seed=9)
    s0 = tr.step_seed()
    a, ctx = head_forward_train(tok, hidden.to(DEV), 2, threshold=thr, dropout_seed=tr.step_seed())
    tr.backward(ctx, torch.cat(ups, 0).to(DEV)); tr.step()
    assert tr.step_seed() != s0 and HeadTrainer(tok, dropout="train", dropout_seed=9).step_seed() == s0
    ev, _ = head_forward_train(tok, hidden.to(DEV), 2, threshold=thr)
    assert torch.isfinite(a.packed).all() and not torch.equal(a.packed, ev.packed)
    tok.inner_encoder.attn_drop_p = 0.1
    with pytest.raises(NotImplementedError):
        HeadTrainer(tok, dropout="train")
    with pytest.raises(NotImplementedError):
        head_forward_train(tok, hidden.to(DEV), 2, threshold=thr, dropout_seed=1)


@pytest.mark.parametrize("low", [torch.bfloat16, torch.float16])
def test_training_step_bf16_runs_and_reduces_loss(low):
    """bf16 throughput mode (and fp16: what the reference's launches without --bf16 train in, train_setokim.py:326) at ViT-ish head dims: a few steps on a fixed
    batch against a fixed linear probe lower the loss."""
    C, N, B = 256, 64, 8
    vc = dict(hidden_size=C, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, image_size=112, patch_size=14)
    tok = SetokTokenizer(vision_tower=vc, mm_vision_select_layer=-2, hidden_dim=C, token_feat_dim=128, min_cluster_num=8,
                         threshold=0.5, nheads=2, dim_feedforward=512).to(device=DEV, dtype=low).eval()
    tr = HeadTrainer(tok, lr=2e-3, dropout="eval")
    g = torch.Generator().manual_seed(0)
    hidden = torch.randn(B * (N + 1), C, generator=g).to(device=DEV, dtype=low)
    losses = []
    for _ in range(6):
        tokens, ctx = head_forward_train(tok, hidden, B)
        target = torch.ones_like(tokens.packed, dtype=torch.float32) * 0.5
        diff = tokens.packed.float() - target
        losses.append(float((diff ** 2).mean()))
        tr.backward(ctx, (2.0 * diff / diff.numel()).to(low))
        tr.step()
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def test_training_step_full_dims_deterministic_bf16():
    """cfg4 head dims (1024 / 2 heads / ff 4096 -> 4096, 576 patches) in bf16 on 6 images: forward + backward twice from the same state give
    bit-identical gradients (no atomics, fixed reduction orders, split-K partials summed in a fixed order), all finite, none identically zero
    except the key biases (whose gradient is exactly 0: softmax is shift-invariant)."""
    C, N, B = 1024, 576, 6
    vc = dict(hidden_size=C, intermediate_size=4096, num_hidden_layers=1, num_attention_heads=16, image_size=336, patch_size=14)
    tok = SetokTokenizer(vision_tower=vc, mm_vision_select_layer=-1, hidden_dim=C, token_feat_dim=4096, min_cluster_num=64,
                         threshold=0.125, nheads=2, dim_feedforward=4096).to(device=DEV, dtype=torch.bfloat16).eval()
    g = torch.Generator().manual_seed(1)
    hidden = torch.randn(B * (N + 1), C, generator=g).to(device=DEV, dtype=torch.bfloat16)
    tokens, ctx = head_forward_train(tok, hidden, B)
    up = (tokens.packed.float() * 1e-2).to(torch.bfloat16)
    g1 = head_backward(tok, ctx, up)
    tokens2, ctx2 = head_forward_train(tok, hidden, B)
    g2 = head_backward(tok, ctx2, up)
    assert torch.equal(tokens.packed, tokens2.packed)
    assert len(g1) == 34
    for n in g1:
        assert torch.equal(g1[n], g2[n]), n
        assert bool(torch.isfinite(g1[n]).all()), n
        if not n.endswith("qkv.bias"):
            assert float(g1[n].abs().max()) > 0, n
    kb = g1["inner_encoder.layers.0.1.qkv.bias"][C:2 * C]
    assert float(kb.abs().max()) <= 1e-2 * float(g1["inner_encoder.layers.0.1.qkv.bias"].abs().max())


def test_training_checkpoint_resume_is_bit_exact(tmp_path):
    """Stop after 2 steps, save (`tokenizer.bin` under the reference's stage-1 key names + the optimiser state), load into a freshly
    initialised tokenizer / trainer, continue: steps 3 and 4 equal the uninterrupted run bit for bit (weights, moments, outputs)."""
    from setok_amd import checkpoint as Ck
    C, N, B = 256, 64, 8
    vc = dict(hidden_size=C, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, image_size=112, patch_size=14)

    def make(seed):
        torch.manual_seed(seed)
        t = SetokTokenizer(vision_tower=vc, mm_vision_select_layer=-2, hidden_dim=C, token_feat_dim=128, min_cluster_num=8,
                           threshold=0.5, nheads=2, dim_feedforward=512).to(device=DEV, dtype=torch.bfloat16).eval()
        return t, HeadTrainer(t, lr=2e-3, weight_decay=0.01, dropout="eval")

    g = torch.Generator().manual_seed(0)
    hidden = torch.randn(B * (N + 1), C, generator=g).to(device=DEV, dtype=torch.bfloat16)

    def step(tok, tr):
        tokens, ctx = head_forward_train(tok, hidden, B)
        diff = tokens.packed.float() - 0.5
        tr.backward(ctx, (2.0 * diff / diff.numel()).to(torch.bfloat16))
        tr.step()
        return tokens.packed.clone()

    tok_a, tr_a = make(0)
    for _ in range(2):
        step(tok_a, tr_a)
    paths = Ck.save_training_checkpoint(tr_a, tmp_path / "checkpoint-2")
    assert all(k.startswith("tokenizer.") for k in torch.load(paths["tokenizer"]))
    tok_b, tr_b = make(123)                                              # different initial weights: everything must come from the files
    Ck.load_training_checkpoint(tr_b, tmp_path / "checkpoint-2")
    assert tr_b.t == 2 and tr_b.wd == 0.01
    for _ in range(2):
        out_a, out_b = step(tok_a, tr_a), step(tok_b, tr_b)
        assert torch.equal(out_a, out_b)
    for n in tr_a.master:
        assert torch.equal(tr_a.master[n], tr_b.master[n]) and torch.equal(tr_a.m[n], tr_b.m[n]) and torch.equal(tr_a.v[n], tr_b.v[n]), n
        assert torch.equal(tr_a.params[n], tr_b.params[n]), n
    with pytest.raises(KeyError):
        bad = Ck.trainer_state(tr_a); bad["exp_avg"].pop(next(iter(bad["exp_avg"])))
        Ck.load_trainer_state(tr_b, bad)
