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
    with pytest.raises(NotImplementedError):
        SetokimLlamaPrefill(dict(kw, num_key_value_heads=2))


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
