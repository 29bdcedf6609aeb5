"""cfg 5 on a real MI355X: the LLM prefill (setok_amd.llama, csrc/llama.hip + setok_linear) through the C ABI against HuggingFace
LlamaForCausalLM's outputs (tests/golden/llama.npz) and torch references of each new op.  `pytest -m gpu`."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import setok_oracle as O

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import setok_amd
    from setok_amd import ops
    from setok_amd.llama import SetokimLlamaPrefill

DEV = "cuda"


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _rand(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def _rel(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-6), (torch.bfloat16, 8e-3), (torch.float16, 8e-3)])
def test_rmsnorm(dt, tol):
    x, w = (_rand(37, 256, seed=1) * 3).to(dt), (1 + 0.1 * _rand(256, seed=2)).to(dt)
    ref = O.llama_rmsnorm(x.float().to(dt), w, 1e-5)
    got = ops.rmsnorm(x.to(DEV), w.to(DEV), 1e-5)
    assert _rel(got, ref.double()) < tol
    if dt == torch.bfloat16:                              # same two roundings as the eager bf16 graph: at most 1 bf16 ulp apart
        assert float((got.cpu().float() - ref.float()).abs().max()) <= 2.0 ** -7 * float(ref.float().abs().max())


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1e-2), (torch.float16, 1e-2)])
def test_rope(dt, tol):
    rows, H, Dh = 50, 3, 32
    qkv = _rand(rows, 3 * H * Dh, seed=3).to(dt)
    pos = torch.randint(0, 3000, (rows,), generator=torch.Generator().manual_seed(4))
    cos, sin = O.llama_rope_tables(pos[None], Dh, 10000.0, dt)
    q = qkv[:, :2 * H * Dh].reshape(rows, 2 * H, Dh)
    ref = (q * cos[0][:, None]) + (O._rotate_half(q) * sin[0][:, None])
    got = ops.rope_(qkv.to(DEV).clone(), pos.to(DEV), H, Dh, 10000.0).cpu()
    assert _rel(got[:, :2 * H * Dh], ref.reshape(rows, -1).double()) < tol
    assert torch.equal(got[:, 2 * H * Dh:], qkv[:, 2 * H * Dh:])            # v untouched


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-6), (torch.bfloat16, 8e-3), (torch.float16, 8e-3)])
def test_swiglu(dt, tol):
    gu = (_rand(33, 2 * 176, seed=5) * 2).to(dt)
    ref = F.silu(gu[:, :176]) * gu[:, 176:]
    assert _rel(ops.swiglu(gu.to(DEV)), ref.double()) < tol


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
def test_swiglu_pairs_equals_swiglu_on_the_deinterleaved_buffer(dt):
    """setok_swiglu_pairs reads (gate_j, up_j) column pairs — the output layout of a Linear with pair-interleaved weight rows; same arithmetic, same bits as
    setok_swiglu on [gate | up]."""
    Fd = 352
    gu = (_rand(77, 2 * Fd, seed=6) * 2).to(dt)
    pairs = torch.stack([gu[:, :Fd], gu[:, Fd:]], dim=2).reshape(77, 2 * Fd).contiguous()
    a, b = ops.swiglu(gu.to(DEV)), ops.swiglu_pairs(pairs.to(DEV))
    assert torch.equal(a, b)
    assert _rel(a, (F.silu(gu[:, :Fd].double()) * gu[:, Fd:].double())) < (2e-6 if dt == torch.float32 else 8e-3)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,Fd,K", [(8192 + 77, 2048, 512), (16990, 11008, 4096 // 8), (4096, 1536, 1024), (300, 512, 256)])
def test_gate_up_linear_with_swiglu_in_its_epilogue(dt, M, Fd, K):
    """LlamaMLP's act_fn(gate_proj(x)) * up_proj(x) (HF modeling_llama.py, reached from setokim_llama.py:130-143) as ONE launch: the ping-pong GEMM with SwiGLU in
    its epilogue (setok_linear_swiglu; round 6).  The epilogue keeps torch's rounding points, so the fused launch, the unfused pair
    (setok_linear on the pair-interleaved weight + setok_swiglu_pairs) and the rows `linear_swiglu` leaves to that pair (behind the last whole 256-row tile; problems
    too small for the persistent kernel) give IDENTICAL bits — and all of it sits within 16-bit rounding of the fp64 formula."""
    x = _rand(M, K, seed=1).to(dt)
    wg, wu = (_rand(Fd, K, seed=2) * K ** -0.5).to(dt), (_rand(Fd, K, seed=3) * K ** -0.5).to(dt)
    wp = ops.interleave_gate_up(wg.to(DEV), wu.to(DEV))
    assert torch.equal(wp[0::2].cpu(), wg) and torch.equal(wp[1::2].cpu(), wu)
    got = ops.linear_swiglu(x.to(DEV), wp)
    unfused = ops.swiglu_pairs(ops.linear(x.to(DEV), wp))
    assert got.dtype == dt and got.shape == (M, Fd) and torch.equal(got, unfused)
    classic = ops.swiglu(ops.linear(x.to(DEV), torch.cat([wg, wu], 0).to(DEV)))            # the [gate | up] layout of rounds 2-5
    assert torch.equal(got, classic)
    g, u = (x.double() @ wg.double().t()).to(dt).double(), (x.double() @ wu.double().t()).to(dt).double()
    ref = F.silu(g).to(dt).double() * u
    assert _rel(got, ref) < (1.5e-2 if dt == torch.bfloat16 else 2e-3)
    sub = ops.linear_swiglu(x[:257].contiguous().to(DEV), wp)                               # a sample alone = the sample inside the batch
    assert torch.equal(sub, got[:257])


def _causal_ref(qkv, km, B, T, H, Dh):
    q, k, v = [t.reshape(B, T, H, Dh).transpose(1, 2).double() for t in qkv.double().reshape(B, T, 3, H * Dh).unbind(2)]
    allow = torch.tril(torch.ones(T, T, dtype=torch.bool))[None, None] & km.bool().reshape(B, 1, 1, T)
    s = (q @ k.transpose(-1, -2)) * Dh ** -0.5
    s = s.masked_fill(~allow, float("-inf"))
    p = torch.softmax(s, -1).nan_to_num(0.0)
    return (p @ v).transpose(1, 2).reshape(B * T, H * Dh)


@pytest.mark.parametrize("dt,tol,H,Dh,T", [(torch.float32, 3e-6, 3, 16, 37), (torch.bfloat16, 1e-2, 2, 128, 300), (torch.bfloat16, 1e-2, 3, 128, 129),
                                            (torch.bfloat16, 1e-2, 1, 128, 31), (torch.bfloat16, 2e-2, 2, 64, 70)])
def test_attention_causal_with_padding(dt, tol, H, Dh, T):
    B = 3
    qkv = _rand(B * T, 3 * H * Dh, seed=6).to(dt)
    km = torch.ones(B, T, dtype=torch.uint8)
    km[1, T - T // 3:] = 0                                 # right padding
    km[2, :T // 4] = 0                                     # left padding: the first queries see no token at all -> zeros
    ref = _causal_ref(qkv, km, B, T, H, Dh)
    got = ops.attention_causal(qkv.to(DEV), km.reshape(-1).to(DEV), B, T, H, Dh, Dh ** -0.5)
    assert _rel(got, ref) < tol
    assert float(got.reshape(B, T, -1)[2, :T // 4].abs().max()) == 0.0
    nomask = ops.attention_causal(qkv.to(DEV), None, B, T, H, Dh, Dh ** -0.5)
    assert _rel(nomask, _causal_ref(qkv, torch.ones(B, T), B, T, H, Dh)) < tol


def _causal_ref_gqa(qkv, km, B, T, H, Hkv, Dh):
    """HF repeat_kv: query head h attends with key / value head h // (H // Hkv)."""
    x = qkv.double().reshape(B, T, (H + 2 * Hkv) * Dh)
    q = x[..., :H * Dh].reshape(B, T, H, Dh).transpose(1, 2)
    k = x[..., H * Dh:(H + Hkv) * Dh].reshape(B, T, Hkv, Dh).transpose(1, 2).repeat_interleave(H // Hkv, dim=1)
    v = x[..., (H + Hkv) * Dh:].reshape(B, T, Hkv, Dh).transpose(1, 2).repeat_interleave(H // Hkv, dim=1)
    allow = torch.tril(torch.ones(T, T, dtype=torch.bool))[None, None] & km.bool().reshape(B, 1, 1, T)
    s = ((q @ k.transpose(-1, -2)) * Dh ** -0.5).masked_fill(~allow, float("-inf"))
    return (torch.softmax(s, -1).nan_to_num(0.0) @ v).transpose(1, 2).reshape(B * T, H * Dh)


@pytest.mark.parametrize("dt,tol,H,Hkv,Dh,T", [(torch.float32, 3e-6, 6, 2, 16, 37), (torch.bfloat16, 1e-2, 4, 2, 128, 200), (torch.bfloat16, 1e-2, 8, 1, 128, 65),
                                                (torch.bfloat16, 2e-2, 4, 2, 64, 70)])
def test_attention_causal_and_rope_grouped_query(dt, tol, H, Hkv, Dh, T):
    """num_key_value_heads < num_attention_heads (Llama-2-70B / Llama-3 / Mistral): rows are [q: H heads | k: Hkv | v: Hkv]; the kernels index the
    shared key / value head instead of materialising HF's repeat_kv copies.  With Hkv == H the entry points are the plain ones (same bits)."""
    B = 2
    W = (H + 2 * Hkv) * Dh
    qkv = _rand(B * T, W, seed=16).to(dt)
    km = torch.ones(B, T, dtype=torch.uint8)
    km[1, :T // 5] = 0
    got = ops.attention_causal(qkv.to(DEV), km.reshape(-1).to(DEV), B, T, H, Dh, Dh ** -0.5, Hkv)
    assert _rel(got, _causal_ref_gqa(qkv, km, B, T, H, Hkv, Dh)) < tol
    pos = torch.randint(0, 3000, (B * T,), generator=torch.Generator().manual_seed(17))
    cos, sin = O.llama_rope_tables(pos[None], Dh, 10000.0, dt)
    qk = qkv[:, :(H + Hkv) * Dh].reshape(B * T, H + Hkv, Dh)
    ref = (qk * cos[0][:, None]) + (O._rotate_half(qk) * sin[0][:, None])
    rot = ops.rope_(qkv.to(DEV).clone(), pos.to(DEV), H, Dh, 10000.0, Hkv).cpu()
    assert _rel(rot[:, :(H + Hkv) * Dh], ref.reshape(B * T, -1).double()) < (4e-5 if dt == torch.float32 else 1e-2)   # (fp32: cos / sin of angles up to 3000 rad)
    assert torch.equal(rot[:, (H + Hkv) * Dh:], qkv[:, (H + Hkv) * Dh:])        # v untouched
    full = _rand(B * T, 3 * H * Dh, seed=18).to(dt).to(DEV)                        # Hkv == H: the plain entry points, bit for bit
    assert torch.equal(ops.attention_causal(full, None, B, T, H, Dh, Dh ** -0.5, H), ops.attention_causal(full, None, B, T, H, Dh, Dh ** -0.5))


def _case(golden_dir, name):
    z = np.load(os.path.join(golden_dir, "llama.npz"))
    kw = {str(k): int(v) for k, v in zip(z[name + ":cfg_keys"], z[name + ":cfg_vals"])}
    lc = O.LlamaConfigLite(**kw)
    seed, B, T, left = [int(v) for v in z[name + ":spec"]]
    sd = O.init_llama_weights(lc, seed=seed)
    x, am, pos = O.llama_inputs(lc, seed, B, T, "left" if left else "right")
    return kw, sd, x, am, pos, _t(z[name + ":hidden"]), _t(z[name + ":logits"])


@pytest.mark.parametrize("name", ["tiny_right", "tiny_left", "dh128", "dh128_left", "gqa_tiny_left", "gqa_dh128", "mqa_dh128_left"])
def test_llama_prefill_fp32_vs_hf(golden_dir, name):
    kw, sd, x, am, pos, hidden, logits = _case(golden_dir, name)
    m = SetokimLlamaPrefill(kw)
    assert not m.load_state_dict(sd, strict=True).missing_keys          # HF LlamaForCausalLM's key names
    m = m.to(DEV).eval()
    lg, _, _ = m(inputs_embeds=x.to(DEV), attention_mask=am.to(DEV), position_ids=pos.to(DEV))
    v = am.bool()
    assert _rel(lg.cpu()[v], logits[v]) < 1e-4
    last, _, _ = m(inputs_embeds=x.to(DEV), attention_mask=am.to(DEV), position_ids=pos.to(DEV), last_token_only=True)
    idx = (am * torch.arange(am.shape[1])[None]).max(1).values
    assert _rel(last.cpu(), logits[torch.arange(am.shape[0]), idx]) < 1e-4


@pytest.mark.parametrize("name", ["dh128", "dh128_left", "gqa_dh128", "mqa_dh128_left"])
def test_llama_prefill_bf16_mfma_attention(golden_dir, name):
    kw, sd, x, am, pos, hidden, logits = _case(golden_dir, name)
    m = SetokimLlamaPrefill(kw)
    m.load_state_dict(sd, strict=True)
    m = m.to(device=DEV, dtype=torch.bfloat16).eval()
    lg, _, _ = m(inputs_embeds=x.to(DEV), attention_mask=am.to(DEV), position_ids=pos.to(DEV))
    v = am.bool()
    assert lg.dtype == torch.bfloat16 and _rel(lg.float().cpu()[v], logits[v]) < 4e-2      # bf16 throughput mode: documented tolerance
    one, _, _ = m(inputs_embeds=x[1:2].to(DEV), attention_mask=am[1:2].to(DEV), position_ids=pos[1:2].to(DEV))
    assert torch.equal(one[0][am[1].bool().to(DEV)], lg[1][am[1].bool().to(DEV)])        # a sequence alone == inside the batch, bit-exact


def test_setokim_forward_splices_then_prefills():
    """input_ids with image placeholders + images -> splice -> LLM -> logits, against the oracle's splice + llama_forward (fp32)."""
    kw = dict(hidden_size=64, intermediate_size=176, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4, vocab_size=100)
    lc = O.LlamaConfigLite(**kw)
    sd = O.init_llama_weights(lc, seed=11)
    ids, am, labels, feats, _ = O.splice_inputs(12, 4, 10, 100, 64)

    class Tower:
        pass

    m = SetokimLlamaPrefill(kw, vision_tower=Tower())
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    m.encode_images = lambda images, **k: [f.to(DEV) for f in feats]
    lg, new_labels, new_am = m(input_ids=ids.to(DEV), attention_mask=am.to(DEV), labels=labels.to(DEV), comp_images=torch.zeros(len(feats), 3, 2, 2))
    _, ram, remb, rlab = O.splice_multimodal(ids, None, am, labels, feats, sd["model.embed_tokens.weight"])
    _, ref = O.llama_forward(sd, lc, remb, ram, None)
    assert torch.equal(new_am.cpu(), ram) and torch.equal(new_labels.cpu(), rlab)
    assert _rel(lg.cpu()[ram.bool()], ref[ram.bool()]) < 1e-4


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-6), (torch.bfloat16, 2e-6), (torch.float16, 2e-6)])
def test_lm_loss_matches_oracle_and_golden(golden_dir, dt, tol):
    """setok_lm_loss against the oracle (setokim_llama.py:145-160) and, in fp32, the reference's own value; the logits are bf16-representable, so
    both dtypes read the same numbers.  Also: a strided logits view, no valid position -> NaN, and the count."""
    z = np.load(os.path.join(golden_dir, "lm_loss.npz"))
    for c in sorted({k.split(":")[0] for k in z.files}):
        seed, B, T, V = (int(v) for v in z[c + ":spec"])
        logits, labels, am = O.lm_loss_inputs(seed, B, T, V, str(z[c + ":padding"]))
        ref = float(O.lm_loss(logits, labels, am))
        Vp = (V + 7) // 8 * 8 + 8                                            # rows with a stride (the kernel takes ld)
        buf = torch.zeros(B, T, Vp, dtype=dt, device=DEV)
        buf[..., :V] = logits.to(dt).to(DEV)
        out = ops.lm_loss(buf[..., :V], labels.to(DEV), None if am is None else am.to(DEV)).cpu()
        assert abs(float(out[0]) - ref) <= tol * abs(ref) + 1e-6
        assert abs(float(out[0]) - float(z[c + ":loss"][0])) <= 2e-6 * abs(ref) + 1e-6
        am1 = torch.ones(B, T, dtype=torch.long) if am is None else am
        n_ref = int(((am1[:, 1:] != 0) & (labels[:, 1:] != -100)).sum())
        assert int(out[1]) == n_ref
    none = ops.lm_loss(buf[..., :V], torch.full((B, T), -100), None).cpu()
    assert bool(torch.isnan(none[0])) and int(none[1]) == 0


def test_prefill_returns_the_lm_loss():
    cfg = dict(vocab_size=96, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=1, rms_norm_eps=1e-5)
    torch.manual_seed(0)
    m = SetokimLlamaPrefill(cfg).to(device=DEV, dtype=torch.bfloat16).eval()
    B, T = 3, 20
    x = torch.randn(B, T, 128).to(device=DEV, dtype=torch.bfloat16)
    labels = torch.randint(0, 96, (B, T)); labels[:, :5] = -100
    am = torch.ones(B, T, dtype=torch.long); am[1, 15:] = 0
    logits, nl, _, loss = m(inputs_embeds=x, attention_mask=am.to(DEV), labels=labels.to(DEV), return_loss=True)
    ref = O.lm_loss(logits.float().cpu(), labels, am)
    assert abs(float(loss) - float(ref)) <= 2e-6 * abs(float(ref))
    with pytest.raises(ValueError):
        m(inputs_embeds=x, attention_mask=am.to(DEV), return_loss=True)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("V", [97, 99, 32003 // 100])
def test_lm_loss_resized_vocabulary_contiguous_rows(dt, V):
    """`initialize_vision_tokenizer` resizes the vocabulary when it adds the image tokens (32002, 32003, ...): contiguous logits rows then start
    at addresses that are not multiples of 16 bytes.  The kernel reads such rows as scalar head | 16-byte vectors | scalar tail."""
    B, T = 3, 9
    g = torch.Generator().manual_seed(V)
    logits = (torch.randn(B, T, V, generator=g) * 3).bfloat16().float()
    labels = torch.randint(0, V, (B, T), generator=g); labels[:, :2] = -100
    am = torch.ones(B, T, dtype=torch.long); am[2, 6:] = 0
    ref = float(O.lm_loss(logits, labels, am))
    dev_logits = logits.to(dt).to(DEV).contiguous()
    assert dev_logits.stride(1) == V
    out = ops.lm_loss(dev_logits, labels.to(DEV), am.to(DEV)).cpu()
    assert abs(float(out[0]) - ref) <= 2e-6 * abs(ref) + 1e-6
    # a view that starts one element into an allocation (misaligned base as well as misaligned stride)
    flat = torch.zeros(B * T * V + 1, dtype=dt, device=DEV)
    flat[1:] = dev_logits.reshape(-1)
    out2 = ops.lm_loss(flat[1:].view(B, T, V), labels.to(DEV), am.to(DEV)).cpu()
    assert abs(float(out2[0]) - ref) <= 2e-6 * abs(ref) + 1e-6


def test_text_only_prefill_validates_ids_like_embed_tokens():
    """the text-only branch is `embed_tokens(input_ids)` (setokim_llama.py:118-128 with images None).  An id the table cannot
    serve — an IMAGE_TOKEN_INDEX although no images came, a leaked TARGET_TOKEN_INDEX, an id >= vocab — raises IndexError like torch's embedding;
    it must never be decoded as an image-token row of a NULL buffer."""
    kw = dict(hidden_size=64, intermediate_size=176, num_hidden_layers=1, num_attention_heads=4, num_key_value_heads=4, vocab_size=100)
    lc = O.LlamaConfigLite(**kw)
    sd = O.init_llama_weights(lc, seed=11)
    m = SetokimLlamaPrefill(kw)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    ids = torch.randint(0, 100, (2, 7), generator=torch.Generator().manual_seed(0))
    lg, _, _ = m(input_ids=ids.to(DEV))
    _, ref = O.llama_forward(sd, lc, sd["model.embed_tokens.weight"][ids], None, None)
    assert _rel(lg.cpu(), ref) < 1e-4
    for bad in (-200, -300, 100, 12345):
        ids2 = ids.clone(); ids2[1, 3] = bad
        with pytest.raises(IndexError, match=r"input_ids\[1, 3\]"):
            m(input_ids=ids2.to(DEV))
    lg2, _, _ = m(input_ids=ids.to(DEV))                                  # the context is intact after the refused calls
    assert torch.equal(lg, lg2)
