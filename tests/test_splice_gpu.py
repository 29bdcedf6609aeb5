"""§8(f) row 1 on a real MI355X: setok_amd.splice_multimodal (setok_splice_lengths / _plan / _rows through the C ABI) against the
outputs of the reference's own prepare_inputs_labels_for_multimodal (tests/golden/splice.npz) — bit-exact: integers and row
copies — and against the CPU oracle at a realistic size.  `pytest -m gpu`."""
import os

import numpy as np
import pytest
import torch

import setok_oracle as O

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import setok_amd
    from setok_amd.tokenizer import RaggedTokens

DEV = "cuda"
NAMES = ["right", "left", "trunc", "trunc_left", "nopad_long"]


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _case(golden_dir, name):
    z = np.load(os.path.join(golden_dir, "splice.npz"))
    seed, B, T, V, D, maxlen, left = [int(v) for v in z[name + ":spec"]]
    ids, am, labels, feats, W = O.splice_inputs(seed, B, T, V, D, pad=not name.startswith("nopad"))
    kw = dict(max_length=None if maxlen < 0 else maxlen, padding_side="left" if left else "right")
    return z, ids, am, labels, feats, W, kw


@pytest.mark.parametrize("name", NAMES)
def test_splice_matches_reference_golden(golden_dir, name):
    z, ids, am, labels, feats, W, kw = _case(golden_dir, name)
    T = ids.shape[1]
    pos = torch.arange(T).expand(ids.shape[0], T).clone()
    ragged = RaggedTokens(torch.cat(feats, 0).to(DEV), [f.shape[0] for f in feats])
    p, a, e, l = setok_amd.splice_multimodal(ids.to(DEV), pos.to(DEV), am.to(DEV), labels.to(DEV), ragged, W.to(DEV), **kw)
    assert torch.equal(e.cpu(), _t(z[f"{name}:full:embeds"]))            # fp32 row copies: bit-exact
    assert torch.equal(p.cpu(), _t(z[f"{name}:full:pos"])) and torch.equal(a.cpu(), _t(z[f"{name}:full:mask"]))
    assert torch.equal(l.cpu(), _t(z[f"{name}:full:labels"]))
    assert p.dtype == pos.dtype and a.dtype == am.dtype and l.dtype == labels.dtype
    # None conventions (setokim_arch.py:341-353) and the list-of-tensors input form
    p, a, e, l = setok_amd.splice_multimodal(ids.to(DEV), None, None, None, [f.to(DEV) for f in feats], W.to(DEV), **kw)
    assert p is None and a is None and l is None and torch.equal(e.cpu(), _t(z[f"{name}:none:embeds"]))


def test_splice_realistic_size_bf16():
    """cfg5-like: 32 sequences of up to 2048 tokens, hidden 4096, one or two images each with ~36 tokens; vs the oracle, bit-exact."""
    B, T, V, D = 32, 2048, 32000, 4096
    g = torch.Generator().manual_seed(9)
    ids = torch.randint(0, V, (B, T), generator=g)
    am = torch.ones(B, T, dtype=torch.bool)
    counts = []
    for b in range(B):
        n = int(torch.randint(T // 2, T + 1, (1,), generator=g)); am[b, n:] = False
        for w in torch.randperm(n, generator=g)[: 1 + b % 2]:
            ids[b, w] = O.IMAGE_TOKEN_INDEX; counts.append(int(torch.randint(24, 56, (1,), generator=g)))
    labels = torch.where(am, ids.clamp_min(0), torch.full_like(ids, O.IGNORE_INDEX))
    feats = [torch.randn(c, D, generator=g).bfloat16() for c in counts]
    W = torch.randn(V, D, generator=g).bfloat16()
    ref = O.splice_multimodal(ids, None, am, labels, feats, W, max_length=2048)
    got = setok_amd.splice_multimodal(ids.to(DEV), None, am.to(DEV), labels.to(DEV), [f.to(DEV) for f in feats], W.to(DEV), max_length=2048)
    assert got[0] is None and torch.equal(got[1].cpu(), ref[1]) and torch.equal(got[3].cpu(), ref[3])
    assert got[2].dtype == torch.bfloat16 and torch.equal(got[2].cpu().view(torch.int16), ref[2].view(torch.int16))
    # properties: kept positions = new lengths; image rows carry IGNORE_INDEX; nothing beyond max_length
    assert got[2].shape[1] <= 2048 and int(got[1].sum()) == int(ref[1].sum())


def test_splice_errors_and_passthrough():
    ids, am, labels, feats, W = O.splice_inputs(7, 4, 10, 30, 8)
    with pytest.raises(IndexError):                                      # one image short: the reference raises at image_features[cur_image_idx]
        setok_amd.splice_multimodal(ids.to(DEV), None, am.to(DEV), labels.to(DEV), [f.to(DEV) for f in feats[:-1]], W.to(DEV))

    class Emb:
        weight = W.to(DEV)

    class M:
        embed_tokens = Emb()

    class Host(setok_amd.SetokimVisionMixin):
        vision_tower = object()
        mm_in_projector = None

        def get_model(self):
            return M()

        def encode_images(self, images, **kw):
            return [f.to(DEV) for f in feats]

    h = Host()
    out = h.prepare_inputs_labels_for_multimodal(ids.to(DEV), None, am.to(DEV), None, labels.to(DEV), torch.zeros(len(feats), 3, 2, 2))
    ref = O.splice_multimodal(ids, None, am, labels, feats, W)
    assert out[0] is None and out[1] is None and out[3] is None
    assert torch.equal(out[2].cpu(), ref[1]) and torch.equal(out[4].cpu(), ref[2]) and torch.equal(out[5].cpu(), ref[3])
    # single-token decode step and images=None pass straight through (setokim_arch.py:218-220)
    one = ids[:, :1].to(DEV)
    assert h.prepare_inputs_labels_for_multimodal(one, None, None, "pkv", None, torch.zeros(1))[0] is one
    assert h.prepare_inputs_labels_for_multimodal(ids.to(DEV), None, None, None, None, None)[4] is None


@pytest.mark.parametrize("as_dict", [True, False])
def test_mixin_honours_truncation_and_left_padding_for_dict_and_attribute_configs(golden_dir, as_dict):
    """setokim_arch.py:304-337: `tokenizer_model_max_length` and `tokenizer_padding_side` come from `self.config`, which hosts hand over
    as an attribute object (HF) or as a plain dict (SetokimLlamaPrefill accepts both); the outputs must equal the reference's own
    golden outputs of the truncated, left-padded case either way."""
    name = "trunc_left"
    z, ids, am, labels, feats, W, kw = _case(golden_dir, name)
    assert kw["padding_side"] == "left" and kw["max_length"] is not None
    cfg = dict(tokenizer_model_max_length=kw["max_length"], tokenizer_padding_side="left")

    class Emb:
        weight = W.to(DEV)

    class M:
        embed_tokens = Emb()

    class Host(setok_amd.SetokimVisionMixin):
        vision_tower = object()
        mm_in_projector = None
        config = cfg if as_dict else type("Cfg", (), cfg)()

        def get_model(self):
            return M()

        def encode_images(self, images, **kw_):
            return [f.to(DEV) for f in feats]

    T = ids.shape[1]
    pos = torch.arange(T).expand(ids.shape[0], T).clone()
    out = Host().prepare_inputs_labels_for_multimodal(ids.to(DEV), pos.to(DEV), am.to(DEV), None, labels.to(DEV), torch.zeros(len(feats), 3, 2, 2))
    assert torch.equal(out[4].cpu(), _t(z[f"{name}:full:embeds"]))
    assert torch.equal(out[1].cpu(), _t(z[f"{name}:full:pos"])) and torch.equal(out[2].cpu(), _t(z[f"{name}:full:mask"]))
    assert torch.equal(out[5].cpu(), _t(z[f"{name}:full:labels"]))
    assert out[4].shape[1] <= kw["max_length"]


def test_ids_outside_the_embedding_table_raise_like_embed_tokens():
    """The reference embeds every kept non-placeholder id with `embed_tokens` (setokim_arch.py:273), which raises IndexError for a negative id
    (e.g. a TARGET_TOKEN_INDEX = -300 that leaked into input_ids) or one >= vocab; masked-out positions are never embedded (:258-259)."""
    ids, am, labels, feats, W = O.splice_inputs(7, 4, 10, 30, 8)
    fd = [f.to(DEV) for f in feats]
    kept = [(b, t) for b in range(ids.shape[0]) for t in range(ids.shape[1]) if am[b, t] and ids[b, t] != O.IMAGE_TOKEN_INDEX]
    dropped = [(b, t) for b in range(ids.shape[0]) for t in range(ids.shape[1]) if not am[b, t]]
    b, t = kept[len(kept) // 2]
    for bad in (O.TARGET_TOKEN_INDEX, -1, W.shape[0], W.shape[0] + 5):
        bad_ids = ids.clone(); bad_ids[b, t] = bad
        with pytest.raises(IndexError, match=rf"input_ids\[{b}, {t}\]"):
            setok_amd.splice_multimodal(bad_ids.to(DEV), None, am.to(DEV), labels.to(DEV), fd, W.to(DEV))
    if dropped:                                                          # an out-of-table id under the mask is harmless, as in the reference
        ok_ids = ids.clone(); ok_ids[dropped[0]] = W.shape[0] + 7
        got = setok_amd.splice_multimodal(ok_ids.to(DEV), None, am.to(DEV), labels.to(DEV), fd, W.to(DEV))
        ref = O.splice_multimodal(ids, None, am, labels, feats, W)
        assert torch.equal(got[2].cpu(), ref[2])
