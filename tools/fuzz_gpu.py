#!/usr/bin/env python
"""Randomised parity sweep on the GPU (beyond the fixed shapes of tests/): python tools/fuzz_gpu.py [seconds] [seed] [bf16|f16]
(the third argument picks the 16-bit build under test: bf16 = libsetok_hip.so, f16 = libsetok_hip_f16.so — round 6).
Linear (all kernels, activations, residual, folded LayerNorm), uniform / ragged / cross attention, the fused clustering against the multi-kernel
form and against itself in other batch positions.  Prints the first failing case and exits non-zero."""
import sys, os, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from setok_amd import ops

DEV = "cuda"
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
LOW = torch.float16 if len(sys.argv) > 3 and sys.argv[3] == "f16" else torch.bfloat16


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def fuzz_linear():
    M = rng.choice([1, 3, 64, 255, 257, 300, 771, 1028, 4100, 9300, 16448, 256 * rng.randint(8, 72), 256 * rng.randint(8, 72) + rng.randint(1, 255), rng.randint(1, 20000)])
    N = rng.choice([64 * rng.randint(1, 48), 256 * rng.randint(1, 16)])      # multiples of 256 with >= 48 tiles: the ping-pong kernel (whole tiles) + the small-tile remainder
    K = 64 * rng.randint(1, 64)
    act = rng.randint(0, 2)
    use_res = rng.random() < 0.5
    g = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
    a = torch.randn(M, K, generator=g).to(LOW).to(DEV)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(LOW).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    r = torch.randn(M, N, generator=g).to(LOW).to(DEV) if use_res else None
    got = ops.linear(a, w, b, r, act=act).float()
    ref = a.float() @ w.float().t() + b
    ref = ref * torch.sigmoid(1.702 * ref) if act == 1 else (F.gelu(ref) if act == 2 else ref)
    ref = ref.to(LOW).float() + (r.float() if use_res else 0)
    e = rel(got, ref)
    assert e < 2e-2, ("linear", M, N, K, act, use_res, e)
    # a slice of the rows through the small kernel: identical bits
    if M > 600:
        lo = rng.randint(0, M - 300)
        sub = ops.linear(a[lo:lo + 257].contiguous(), w, b, None if r is None else r[lo:lo + 257].contiguous(), act=act)
        assert torch.equal(sub, got[lo:lo + 257].to(sub.dtype)), ("linear rows differ between kernels", M, N, K, act, use_res, lo)
    return ("linear", M, N, K, act, use_res)


def fuzz_linear_ln():
    M = rng.choice([5, 257, 771, 4100, 16448, 256 * rng.randint(8, 72), 256 * rng.randint(8, 72) + rng.randint(1, 255), rng.randint(1, 12000)])
    N = rng.choice([64 * rng.randint(1, 48), 256 * rng.randint(1, 16)])
    K = 64 * rng.randint(1, 32)
    act = rng.randint(0, 2)
    g = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
    x = (torch.randn(M, K, generator=g) * (0.5 + 2 * torch.rand(M, 1, generator=g)) + torch.randn(M, 1, generator=g)).to(LOW).to(DEV)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(LOW).to(DEV)
    gamma, beta, bias = (1 + 0.2 * torch.randn(K, generator=g)).to(DEV), (0.1 * torch.randn(K, generator=g)).to(DEV), (0.1 * torch.randn(N, generator=g)).to(DEV)
    st = ops.row_stats(x, 1e-5)
    got = ops.linear_ln(x, ops.ln_fold(w, gamma, beta, bias), st, act=act).float()
    y = F.layer_norm(x.double(), (K,), gamma.double(), beta.double(), 1e-5) @ w.double().t() + bias.double()
    y = y * torch.sigmoid(1.702 * y) if act == 1 else (F.gelu(y) if act == 2 else y)
    e = rel(got, y)
    assert e < 1.5e-2, ("linear_ln", M, N, K, act, e)
    if M > 600:
        lo = rng.randint(0, M - 300)
        sub = ops.linear_ln(x[lo:lo + 257].contiguous(), ops.ln_fold(w, gamma, beta, bias), st[lo:lo + 257].contiguous(), act=act)
        assert torch.equal(sub, got[lo:lo + 257].to(sub.dtype)), ("linear_ln rows differ between kernels", M, N, K, act, lo)
    return ("linear_ln", M, N, K, act)


def attn_ref(q, k, v, scale):
    return torch.softmax(q.double() @ k.double().transpose(-1, -2) * scale, -1) @ v.double()


def fuzz_attention():
    Dh = rng.choice([64, 64, 64, 48, 16, 96, 512])
    H = rng.choice([1, 2, 4, 12, 16]) if Dh != 512 else 2
    T = rng.choice([1, 17, 32, 33, 64, 197, 256, 257, 300, 324, 577, rng.randint(1, 608)])
    n = rng.randint(1, 4)
    g = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
    qkv = torch.randn(n * T, 3 * H * Dh, generator=g).to(LOW).to(DEV)
    got = ops.attention(qkv, H, Dh, Dh ** -0.5, seg_len=T).float()
    q, k, v = qkv.float().reshape(n, T, 3, H, Dh).permute(2, 0, 3, 1, 4)
    ref = attn_ref(q, k, v, Dh ** -0.5).transpose(1, 2).reshape(n * T, H * Dh)
    e = rel(got, ref)
    assert e < 1.5e-2, ("attention", T, H, Dh, n, e)
    return ("attention", T, H, Dh, n)


def fuzz_ragged_attention():
    Dh = rng.choice([64, 512, 16, 48])
    H = 2 if Dh == 512 else rng.choice([1, 2, 4])
    lens = [rng.choice([1, 2, 3, 7, 31, 32, 33, 64, 65, rng.randint(1, 300)]) for _ in range(rng.randint(1, 12))]
    offs = [0]
    for L in lens:
        offs.append(offs[-1] + L)
    g = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
    qkv = torch.randn(offs[-1], 3 * H * Dh, generator=g).to(LOW).to(DEV)
    so = torch.tensor(offs, dtype=torch.int32, device=DEV)
    got = ops.attention(qkv, H, Dh, Dh ** -0.5, seg_len=max(lens), seg_offsets=so, n_segs=len(lens)).float()
    for s0, s1 in zip(offs[:-1], offs[1:]):
        q, k, v = qkv[s0:s1].float().reshape(s1 - s0, 3, H, Dh).permute(1, 2, 0, 3)
        ref = attn_ref(q, k, v, Dh ** -0.5).transpose(0, 1).reshape(s1 - s0, H * Dh)
        e = rel(got[s0:s1], ref)
        assert e < 1.5e-2, ("ragged attention", Dh, H, lens, e)
    return ("ragged attention", Dh, H, len(lens))


def fuzz_cross_attention():
    H, Dh = rng.choice([(12, 64), (4, 64), (2, 64), (3, 96), (4, 16)])
    q_len = rng.choice([1, 5, 25, 32, 33, 64, 256, 257, 324])
    lens = [rng.choice([1, 2, 8, 9, 31, 32, 33, 64, 65, rng.randint(1, 200)]) for _ in range(rng.randint(1, 5))]
    offs = [0]
    for L in lens:
        offs.append(offs[-1] + L)
    C = H * Dh
    g = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
    q = torch.randn(len(lens) * q_len, C, generator=g).to(LOW).to(DEV)
    kv = torch.randn(offs[-1], 2 * C, generator=g).to(LOW).to(DEV)
    so = torch.tensor(offs, dtype=torch.int32, device=DEV)
    got = ops.cross_attention(q, kv[:, :C], kv[:, C:], H, Dh, Dh ** -0.5, q_len, so, len(lens), max(lens)).float()
    for i, (s0, s1) in enumerate(zip(offs[:-1], offs[1:])):
        qq = q[i * q_len:(i + 1) * q_len].float().reshape(q_len, H, Dh).transpose(0, 1)
        kk = kv[s0:s1, :C].float().reshape(s1 - s0, H, Dh).transpose(0, 1)
        vv = kv[s0:s1, C:].float().reshape(s1 - s0, H, Dh).transpose(0, 1)
        ref = attn_ref(qq, kk, vv, Dh ** -0.5).transpose(0, 1).reshape(q_len, C)
        e = rel(got[i * q_len:(i + 1) * q_len], ref)
        assert e < 1.5e-2, ("cross attention", H, Dh, q_len, lens, e)
    return ("cross attention", H, Dh, q_len, len(lens))


def fuzz_cluster():
    N = rng.choice([256, 256, 196, 64, 144, rng.randint(2, 256), 576, 324, rng.randint(257, 576)])     # > 256: the strip kernel (workgroups of an image exchange rho / scores)
    C = 64 * rng.randint(1, 16)
    B = rng.randint(1, 9) if N <= 256 else rng.choice([1, 2, 3, 60, 130])     # (> 51 images: more items than workgroups)
    k = rng.randint(1, min(N, 96))
    mcn = rng.randint(1, min(N, 64))
    thr = rng.choice([0.5, 0.12, 0.13, 1e9, 0.0])
    g = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
    m = rng.randint(1, 16)
    cent = torch.randn(B, m, C, generator=g) * 2
    lab = torch.randint(0, m, (B, N), generator=g)
    x = (torch.gather(cent, 1, lab[..., None].expand(B, N, C)) + rng.choice([0.05, 0.5, 1.0]) * torch.randn(B, N, C, generator=g)).to(LOW).to(DEV).reshape(B * N, C)
    noise = torch.rand(B, N, generator=g).to(DEV) if rng.random() < 0.5 else None
    mask = (torch.rand(B, N, generator=g) > 0.2).float().to(DEV) if rng.random() < 0.3 else None
    a = ops.cluster_dpc_knn(x, B, N, k, thr, mcn, noise, mask)
    # determinism, and an image's result does not depend on its batch position
    b = ops.cluster_dpc_knn(x, B, N, k, thr, mcn, noise, mask)
    assert all(torch.equal(p, q) for p, q in zip(a, b)), ("cluster not deterministic", N, C, B, k, mcn, thr)
    i = rng.randint(0, B - 1)
    one = ops.cluster_dpc_knn(x[i * N:(i + 1) * N].contiguous(), 1, N, k, thr, mcn, None if noise is None else noise[i:i + 1].contiguous(),
                              None if mask is None else mask[i:i + 1].contiguous())
    assert torch.equal(one[0].reshape(-1), a[0].reshape(B, N)[i]) and torch.equal(one[1].reshape(-1), a[1].reshape(B, N)[i]) and int(one[3][0]) == int(a[3][i]), \
        ("cluster depends on the batch", N, C, B, k, mcn, thr, i)
    idx, score, down, counts = a
    cnt = counts.cpu().tolist()
    assert all(1 <= c <= N for c in cnt), ("counts", cnt)
    assert torch.isfinite(score).all(), ("score not finite", N, C, B, k, mcn, thr)
    idxc = idx.reshape(B, N).cpu(); downc = down.reshape(B, -1).cpu()
    for bi in range(B):
        cs = downc[bi, :cnt[bi]]
        assert bool((idxc[bi] >= 0).all()) and bool((idxc[bi] < cnt[bi]).all()), ("labels out of range", N, C, B, k, mcn, thr)
        assert bool((idxc[bi][cs] == torch.arange(cnt[bi])).all()), ("a centre does not own itself", N, C, B, k, mcn, thr)
    return ("cluster", N, C, B, k, mcn, thr, noise is not None, mask is not None)


FUZZERS = [fuzz_linear, fuzz_linear, fuzz_linear_ln, fuzz_attention, fuzz_attention, fuzz_ragged_attention, fuzz_cross_attention, fuzz_cluster, fuzz_cluster]
n = 0
last = None
try:
    while time.time() < t_end:
        last = rng.choice(FUZZERS)()
        n += 1
except AssertionError as e:
    print("FAIL after", n, "cases:", e.args[0] if e.args else e, flush=True)
    sys.exit(1)
print(f"ok: {n} random cases, last {last}")
