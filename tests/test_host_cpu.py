"""Host-side contracts that need no GPU: the mirrors carry exactly the reference's / HuggingFace's state-dict key names and shapes, so
checkpoints load unchanged; constructor errors mirror the configurations the reference itself cannot run."""
import pytest
import torch

import setok_oracle as O


def test_llama_mirror_has_hf_state_dict_keys():
    from transformers import LlamaConfig, LlamaForCausalLM
    from setok_amd.llama import SetokimLlamaPrefill
    kw = dict(hidden_size=64, intermediate_size=176, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4, vocab_size=100)
    hf = LlamaForCausalLM(LlamaConfig(**kw, attention_bias=False, mlp_bias=False, tie_word_embeddings=False))
    mine = SetokimLlamaPrefill(kw)
    a, b = hf.state_dict(), mine.state_dict()
    assert set(a) == set(b)
    assert all(tuple(a[k].shape) == tuple(b[k].shape) for k in a)
    assert set(O.init_llama_weights(O.LlamaConfigLite(**kw))) == set(a)
    kw2 = dict(kw, num_key_value_heads=2)                                       # grouped-query attention: k_proj / v_proj of Hkv heads, the same key names
    hf2 = LlamaForCausalLM(LlamaConfig(**kw2, attention_bias=False, mlp_bias=False, tie_word_embeddings=False)).state_dict()
    mine2 = SetokimLlamaPrefill(kw2).state_dict()
    assert set(hf2) == set(mine2) and all(tuple(hf2[k].shape) == tuple(mine2[k].shape) for k in hf2)
    with pytest.raises(ValueError):
        SetokimLlamaPrefill(dict(kw, num_key_value_heads=3))                   # 4 query heads cannot share 3 key / value heads


def test_detokenizer_mirror_keys_and_ctor_errors():
    from setok_amd import SetokDeTokenizer
    dc = O.DetokConfig(token_feat_dim=96, hidden_dim=64, image_size=70, decoder_embed_dim=64, decoder_nheads=4, decoder_depth=2,
                       num_hidden_layers=4, mapper_hidden=64, mapper_heads=4, mapper_intermediate=128)
    det = SetokDeTokenizer(token_feat_dim=96, hidden_dim=64, image_size=70, decoder_embed_dim=64, decoder_nheads=4, decoder_depth=2,
                           num_hidden_layers=4, feature_mapper_path_or_name=dict(hidden_size=64, num_attention_heads=4, intermediate_size=128))
    want = set(O.init_detok_weights(dc)) | {"position_embedding.inv_freq"}
    assert set(det.state_dict()) == want
    assert det.num_mask_token == 25 and det.mask_tokens.shape == (1, 25, 64)
    with pytest.raises(ValueError):
        SetokDeTokenizer()                                       # hidden_dim 4096 into LayerNorm(768): the reference's own default cannot run
    with pytest.raises(ValueError):
        SetokDeTokenizer(hidden_dim=768, decoder_embed_dim=4096)  # x + pos_emb cannot broadcast 768 -> 4096 channels


def test_tokenizer_mirror_keys_match_oracle_init():
    from setok_amd import SetokTokenizer
    hc = O.HeadConfig(hidden_dim=64, token_feat_dim=96, min_cluster_num=8, threshold=0.5, nheads=2, dim_feedforward=128)
    vc = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, image_size=112, patch_size=14)
    tok = SetokTokenizer(vision_tower=vc, hidden_dim=64, token_feat_dim=96, min_cluster_num=8, nheads=2, dim_feedforward=128)
    head = {k for k in tok.state_dict() if not k.startswith(("image_feature_encoder", "position_embedding"))}
    aliases = {k for k in head if ".layers." in k and k.split(".layers.")[1].split(".")[1] == "0"}      # `layers.{i}.0.*` alias norm1 (module.py:87-88)
    assert head - aliases == set(O.init_head_weights(hc)) and len(aliases) == 8
    from setok_amd.training import HEAD_MODULES
    trainable = {n for n, _ in tok.named_parameters() if n.split(".")[0] in HEAD_MODULES}
    assert len(trainable) == 34                                  # the reference's 34 distinct head parameters (norm1 shared by the sub-layers)


def test_splice_constants_match_reference_constants():
    from setok_amd import arch
    assert (arch.IGNORE_INDEX, arch.IMAGE_TOKEN_INDEX, arch.TARGET_TOKEN_INDEX) == (O.IGNORE_INDEX, O.IMAGE_TOKEN_INDEX, O.TARGET_TOKEN_INDEX) == (-100, -200, -300)


# ---------------------------------------------------------------------------------------------------------------------------------------
# checkpoint hand-over with the reference's key names (setokim_arch.py:94-99, :115-120; setokim_trainer.py:234-251)
# ---------------------------------------------------------------------------------------------------------------------------------------
def _small_tok(seed):
    from setok_amd import SetokTokenizer
    torch.manual_seed(seed)
    vc = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, image_size=112, patch_size=14)
    tok = SetokTokenizer(vision_tower=vc, hidden_dim=64, token_feat_dim=96, min_cluster_num=8, nheads=2, dim_feedforward=128)
    with torch.no_grad():
        for p in tok.parameters():
            p.copy_(torch.randn_like(p))
    return tok


def test_select_by_keyword_is_the_references_get_w():
    from setok_amd.checkpoint import select_by_keyword
    w = {"tokenizer.out.weight": torch.ones(1), "model.tokenizer.out.bias": torch.zeros(1), "detokenizer.decoder_norm.weight": torch.full((1,), 2.0),
         "loss.logit_scale": torch.full((1,), 3.0)}
    got = select_by_keyword(w, "tokenizer")
    # keys CONTAINING the keyword, cut after its first occurrence — detokenizer keys come along (reference quirk, strict=False drops them)
    assert set(got) == {"out.weight", "out.bias", "decoder_norm.weight"}
    assert got["decoder_norm.weight"].item() == 2.0
    with pytest.raises(IndexError):
        select_by_keyword({"tokenizer_scale": torch.ones(1)}, "tokenizer")          # contains the keyword, not `keyword.`: the reference raises too


def test_tokenizer_checkpoint_round_trip(tmp_path):
    from setok_amd import checkpoint as C
    a, b = _small_tok(0), _small_tok(1)
    sd = C.save_tokenizer_checkpoint(a, tmp_path / "setok.bin")
    assert all(k.startswith("tokenizer.") for k in sd) and not any("image_feature_encoder" in k for k in sd)
    assert "tokenizer.inner_encoder.layers.0.0.weight" in sd and "tokenizer.out.weight" in sd
    missing, unexpected = C.load_pretrained_tokenizer(b, tmp_path / "setok.bin")
    assert unexpected == [] and all(k.startswith("image_feature_encoder.") for k in missing)        # the frozen tower is not in the file
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.equal(p, q) == (not n.startswith("image_feature_encoder.")), n
    full = C.tokenizer_checkpoint(a, include_tower=True)
    assert any(k.startswith("tokenizer.image_feature_encoder.") for k in full)
    assert C.load_pretrained_tokenizer(b, full).missing_keys == []
    assert all(torch.equal(p, q) for p, q in zip(a.parameters(), b.parameters()))


def test_adapter_checkpoint_round_trip(tmp_path):
    from setok_amd import build_vision_projector, checkpoint as C

    class Inner(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.mm_in_projector = build_vision_projector("mlp2x_gelu", 32, 48)
            self.embed_tokens = torch.nn.Embedding(10, 48)

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.model = Inner()
            self.lm_head = torch.nn.Linear(48, 10, bias=False)

    torch.manual_seed(0)
    m = Model()
    path = C.save_adapter_checkpoint(m, tmp_path / "checkpoint-10")
    assert path.endswith("mm_projector.bin")
    sd = torch.load(path)
    assert set(sd) == {f"model.mm_in_projector.{i}.{w}" for i in (0, 2) for w in ("weight", "bias")}      # names untouched, nothing else saved
    torch.manual_seed(1)
    p = build_vision_projector("mlp2x_gelu", 32, 48)
    res = C.load_pretrained_projector(p, path)
    assert res.missing_keys == [] and res.unexpected_keys == []
    assert all(torch.equal(x, y) for x, y in zip(p.parameters(), m.model.mm_in_projector.parameters()))


def test_packed_weight_caches_are_dropped_by_loads_through_a_parent_module():
    """The compute-ready copies (fused q|k|v, fp32 biases) of the tower, the Blocks and the LLM must not survive a state-dict load that
    arrives through a PARENT module (`SetokTokenizer.load_state_dict`, `load_pretrained_tokenizer`): nn.Module recurses with
    `_load_from_state_dict`, so a child's own `load_state_dict` override never runs.  Also in-place updates (optimiser step)."""
    tok = _small_tok(0)
    tower, blk = tok.image_feature_encoder, tok.inner_encoder
    pk_t, pk_b = tower._pack(), blk._pack()
    assert tower._packed and blk._packed
    w_before = pk_t["layers"][0]["wqkv"].clone()
    sd = {k: (v + 1.0 if v.is_floating_point() else v) for k, v in tok.state_dict().items()}
    tok.load_state_dict(sd)                                        # through the parent
    assert not tower._packed and not blk._packed
    assert torch.equal(tower._pack()["layers"][0]["wqkv"], w_before + 1.0)
    assert torch.equal(blk._pack()["b1"], tok.inner_encoder.mlp.fc1.bias.detach().float())
    # in-place update of ONE tensor that is not the tensor the old key looked at
    k0 = blk._pack()["key"]
    with torch.no_grad():
        blk.norm2.bias.add_(1.0)
    assert blk._pack()["key"] != k0 and torch.equal(blk._pack()["n2"][1], blk.norm2.bias.detach().float())
    from setok_amd.llama import SetokimLlamaPrefill
    kw = dict(hidden_size=64, intermediate_size=176, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4, vocab_size=100)
    m = SetokimLlamaPrefill(kw)
    m.model._pack()
    assert m.model._packed
    m.load_state_dict({k: v * 2 for k, v in m.state_dict().items()})
    assert not m.model._packed
    assert torch.equal(m.model._pack()["layers"][1]["wd"], m.model.layers[1].mlp.down_proj.weight.detach())


def test_config_get_reads_dicts_and_attribute_objects():
    from setok_amd.arch import config_get
    assert config_get(dict(tokenizer_padding_side="left"), "tokenizer_padding_side", "right") == "left"
    assert config_get(type("C", (), dict(tokenizer_model_max_length=7))(), "tokenizer_model_max_length") == 7
    assert config_get(None, "x", 3) == 3 and config_get({}, "x", 4) == 4


def test_head_trainer_says_that_it_trains_without_dropout():
    """The reference trains the head with proj_drop = 0.2 active (module.py:29-73); the hand-written training step runs eval-mode arithmetic
    and must say so instead of silently optimising another objective."""
    from setok_amd.training import HeadTrainer
    tok = _small_tok(0)                                            # built with the reference's default proj_drop = 0.2
    assert tok.inner_encoder.proj_drop_p == 0.2 and tok.inter_encoder.attn_drop_p == 0.0
    with pytest.warns(UserWarning, match="without dropout"):
        HeadTrainer(tok)
    with pytest.raises(NotImplementedError):
        HeadTrainer(tok, dropout="error")
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        HeadTrainer(tok, dropout="eval")                           # accepted explicitly: silent
        assert HeadTrainer(tok, dropout="train", dropout_seed=3).step_seed() == HeadTrainer(tok, dropout="train", dropout_seed=3).step_seed()   # the masks ARE applied: nothing to warn about
    with pytest.raises(ValueError):
        HeadTrainer(tok, dropout="maybe")


def test_tokenizer_copies_do_not_share_the_library_context():
    """copy.deepcopy / pickle of the module (EMA copies, torch.save of a whole module) must not duplicate the C handle of the encode context."""
    import copy
    import pickle
    tok = _small_tok(0)
    tok.__dict__["_ctx"] = ("key", object())                       # stands in for a built context (no GPU here)
    tok.__dict__["_ctx_params"] = list(tok.parameters())
    for clone in (copy.deepcopy(tok), pickle.loads(pickle.dumps(tok))):
        assert "_ctx" not in clone.__dict__ and "_ctx_params" not in clone.__dict__
        assert torch.equal(clone.out.weight, tok.out.weight)
    assert "_ctx" in tok.__dict__


def test_llama_prefill_refuses_config_fields_it_would_silently_ignore():
    """only num_key_value_heads and rope_theta are read beyond the plain Llama sizes; a config that changes the arithmetic in any
    other way (rope_scaling, an explicit head_dim, projection biases) must not load and give wrong logits."""
    from setok_amd.llama import SetokimLlamaPrefill
    base = dict(vocab_size=64, hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=4, num_key_value_heads=2,
                rms_norm_eps=1e-5, rope_theta=10000.0)
    SetokimLlamaPrefill(dict(base))                                                          # plain grouped-query Llama: fine
    SetokimLlamaPrefill(dict(base, rope_scaling=None, head_dim=8, attention_bias=False, mlp_bias=False, sliding_window=None))
    SetokimLlamaPrefill(dict(base, rope_scaling={"rope_type": "default"}))
    for bad in (dict(rope_scaling={"rope_type": "llama3", "factor": 8.0}), dict(rope_scaling={"type": "linear", "factor": 2.0}),
                dict(head_dim=16), dict(attention_bias=True), dict(mlp_bias=True)):
        with pytest.raises(NotImplementedError):
            SetokimLlamaPrefill(dict(base, **bad))

    # newer HF configs keep the rotary base INSIDE rope_parameters; it must be read from there, and a conflict refused
    no_top = {k: v for k, v in base.items() if k != "rope_theta"}
    assert SetokimLlamaPrefill(dict(no_top, rope_parameters={"rope_type": "default", "rope_theta": 500000.0})).model.rope_theta == 500000.0
    assert SetokimLlamaPrefill(dict(base, rope_parameters={"rope_type": "default", "rope_theta": 10000.0})).model.rope_theta == 10000.0
    assert SetokimLlamaPrefill(dict(no_top)).model.rope_theta == 10000.0
    with pytest.raises(NotImplementedError):
        SetokimLlamaPrefill(dict(base, rope_parameters={"rope_type": "default", "rope_theta": 500000.0}))

    class Cfg:                                                                               # attribute-style configs (HF) are read the same way
        pass
    c = Cfg()
    for k, v in dict(base, rope_scaling={"rope_type": "yarn", "factor": 4.0}).items():
        setattr(c, k, v)
    with pytest.raises(NotImplementedError):
        SetokimLlamaPrefill(c)


def test_no_grad_warning_is_per_module_instance_and_words_inputs_separately():
    """keyed by name alone the first warning silenced every other tower of the process, and an image that requires a gradient
    produced (or swallowed) the message about parameters."""
    import warnings
    from setok_amd import autograd
    a, b = torch.nn.Linear(2, 2), torch.nn.Linear(2, 2)
    x = torch.zeros(1, 2, requires_grad=True)
    why = "there is no backward pass"
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        autograd.warn_no_grad_once("Tower", a.parameters(), why, owner=a)
        autograd.warn_no_grad_once("Tower", a.parameters(), why, owner=a)                    # same instance: once
        autograd.warn_no_grad_once("Tower", b.parameters(), why, owner=b)                    # another instance: its own warning
        autograd.warn_no_grad_once("Tower", [p.detach() for p in b.parameters()], why, owner=object(), inputs=[x])
    msgs = [str(m.message) for m in w]
    assert len(msgs) == 3
    assert sum("a parameter requires one" in m for m in msgs) == 2 and sum("an INPUT requires a gradient" in m for m in msgs) == 1
    with warnings.catch_warnings(record=True) as w, torch.no_grad():
        warnings.simplefilter("always")
        autograd.warn_no_grad_once("Tower", torch.nn.Linear(2, 2).parameters(), why, owner=object())
    assert len(w) == 0                                                                       # gradients disabled: nothing to say
