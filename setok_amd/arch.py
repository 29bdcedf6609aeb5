"""`encode_images` — the drop-in boundary of the Setokim pipeline (src/model/setokim_arch.py:206-211) — and the step right
after it, `prepare_inputs_labels_for_multimodal` (setokim_arch.py:213-355)."""
from __future__ import annotations

from typing import Optional

import torch

from . import autograd, ops
from .tokenizer import RaggedTokens

IGNORE_INDEX = -100               # src/constants.py:7
IMAGE_TOKEN_INDEX = -200          # src/constants.py:8
TARGET_TOKEN_INDEX = -300         # src/constants.py:15


def config_get(cfg, name: str, default=None):
    """One accessor for the model config, which callers hand over either as an attribute object (HF `PretrainedConfig`, the
    reference's case) or as a plain dict (`SetokimLlamaPrefill` accepts both): `getattr` on a dict would silently return the default
    and drop `tokenizer_model_max_length` / `tokenizer_padding_side` (setokim_arch.py:311-337)."""
    if cfg is None:
        return default
    if isinstance(cfg, dict):
        return cfg.get(name, default)
    return getattr(cfg, name, default)


def encode_images(vision_tower, mm_in_projector, images, **tower_kwargs):
    """image_features, _, _ = vision_tower(images); image_features = mm_in_projector(image_features).

    Differentiable where the reference trains through it (setok_amd/autograd.py): the projector's parameters, and the tokenizer's head when
    its parameters require a gradient; a frozen tokenizer runs the single-call inference path.

    Returns a RaggedTokens: `feats[i]` is image i's (L_i, hidden) token matrix, which is
    what prepare_inputs_labels_for_multimodal indexes per image (setokim_arch.py:265-293)."""
    image_features, _, _ = vision_tower(images, **tower_kwargs)            # setokim_arch.py:207
    return mm_in_projector(image_features)                                  # :210 (4-D flatten branch :208-209 never taken)


def splice_multimodal(input_ids, position_ids, attention_mask, labels, image_features, embed_weight,
                      max_length: Optional[int] = None, padding_side: str = "right"):
    """The data path of prepare_inputs_labels_for_multimodal after encode_images (setokim_arch.py:241-353) on the device:
    `image_features` is the RaggedTokens `encode_images` returned (or a list of (L_i, D) tensors), `embed_weight` the LLM's
    embedding table (`get_model().embed_tokens.weight`).  Returns (position_ids, attention_mask, inputs_embeds, labels) with the
    reference's None conventions (:341-353).  One host read of B ints (the new lengths) sizes the outputs.

    `inputs_embeds` carries a grad_fn when the image features do (the projector is being trained) or `embed_weight` requires a gradient:
    the backward pass routes d inputs_embeds rows back to the packed image tokens / the embedding table (autograd.SpliceRowsFn)."""
    if isinstance(image_features, (list, tuple)):
        image_features = RaggedTokens(torch.cat(list(image_features), 0) if len(image_features) else embed_weight.new_zeros((0, embed_weight.shape[1])),
                                      [t.shape[0] for t in image_features])
    if not isinstance(image_features, RaggedTokens):                     # dense (n_images, L, D)
        n, L, _ = image_features.shape
        image_features = RaggedTokens(image_features.reshape(n * L, -1), [L] * n)
    dev = embed_weight.device
    B, T = input_ids.shape
    packed = image_features.packed.to(device=dev, dtype=embed_weight.dtype).contiguous()        # (torch's own cast / copy: differentiable)
    n_images = len(image_features)
    img_offsets = torch.from_numpy(image_features.offsets.astype("int32")).to(dev)
    ids = input_ids.to(device=dev, dtype=torch.int64).contiguous()
    am8 = None if attention_mask is None else attention_mask.to(device=dev).bool().to(torch.uint8).contiguous()     # :252-253
    lab = None if labels is None else labels.to(device=dev, dtype=torch.int64).contiguous()
    with torch.no_grad():
        plan = _splice_plan(input_ids, position_ids, attention_mask, labels, image_features, embed_weight, ids, am8, lab, img_offsets, n_images,
                            max_length, padding_side)
    src, new_labels, new_mask, new_pos = plan
    if autograd.grad_needed(packed, embed_weight):
        embeds = autograd.SpliceRowsFn.apply(packed, embed_weight, src)
    else:
        embeds = ops.splice_rows(src, embed_weight.detach().contiguous(), packed.detach() if packed.shape[0] else None)
    return new_pos, new_mask, embeds, new_labels


def _splice_plan(input_ids, position_ids, attention_mask, labels, image_features, embed_weight, ids, am8, lab, img_offsets, n_images,
                 max_length, padding_side):
    """Integer bookkeeping of the splice (lengths, per-position sources, labels / mask / position ids): no gradient flows through it."""
    B, T = input_ids.shape
    seq_len, img_start, status = ops.splice_lengths(ids, am8, img_offsets, n_images, IMAGE_TOKEN_INDEX, max_length or 0,
                                                    vocab=embed_weight.shape[0])
    host = torch.cat([status, seq_len]).cpu()                           # the one synchronisation
    if int(host[0]) != 0:
        raise IndexError(f"the batch needs {int(host[1])} images but only {n_images} were encoded "
                         "(every placeholder, and every sequence without one, consumes an image: setokim_arch.py:264-271,290)")
    if int(host[2]) != 0:                                               # the reference's embed_tokens raises here (setokim_arch.py:273)
        b, t = divmod(int(host[3]), T)
        raise IndexError(f"index out of range in self: input_ids[{b}, {t}] = {int(input_ids[b, t])} is neither IMAGE_TOKEN_INDEX "
                         f"({IMAGE_TOKEN_INDEX}) nor a row of the {embed_weight.shape[0]}-row embedding table")
    max_len = int(host[4:].max())
    src, new_labels, new_mask, new_pos = ops.splice_plan(ids, am8, lab, img_offsets, seq_len, img_start, max_len, padding_side == "left",
                                                         IMAGE_TOKEN_INDEX, IGNORE_INDEX, TARGET_TOKEN_INDEX,
                                                         want_mask=attention_mask is not None, want_pos=position_ids is not None)
    if new_mask is not None:
        new_mask = new_mask.to(attention_mask.dtype)                     # :346-349
    if new_pos is not None:
        new_pos = new_pos.to(position_ids.dtype)
    if new_labels is not None:
        new_labels = new_labels.to(labels.dtype)
    return src, new_labels, new_mask, new_pos


class SetokimVisionMixin:
    """Methods a LLaVA-style meta-model expects (setokim_arch.py:176-211)."""

    def get_vision_tower(self):
        return self.vision_tower

    def get_input_projector(self):
        return self.mm_in_projector

    def encode_images(self, images, **kw):
        return encode_images(self.get_vision_tower(), self.get_input_projector(), images, **kw)

    def prepare_inputs_labels_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values, labels, images,
                                             image_sizes=None):
        """setokim_arch.py:213-355, same signature and return tuple.  Needs `get_model().embed_tokens` and `config`."""
        vision_tower = self.get_vision_tower()
        if vision_tower is None or images is None or input_ids.shape[1] == 1:                         # :218-220
            return input_ids, position_ids, attention_mask, past_key_values, None, labels
        if type(images) is list or images.ndim == 5:                                                   # :222-225
            images = torch.stack([image for image in images], dim=0)
        image_features = self.encode_images(images)
        self._last_features = image_features                                                           # kept for callers that report token counts
        cfg = getattr(self, "config", None)
        pos, am, embeds, new_labels = splice_multimodal(
            input_ids, position_ids, attention_mask, labels, image_features, self.get_model().embed_tokens.weight,
            max_length=config_get(cfg, "tokenizer_model_max_length", None),                             # :311-314
            padding_side=config_get(cfg, "tokenizer_padding_side", "right"))                            # :323
        return None, pos, am, past_key_values, embeds, new_labels
