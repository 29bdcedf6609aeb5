"""`encode_images` — the drop-in boundary of the Setokim pipeline (src/model/setokim_arch.py:206-211)."""
from __future__ import annotations

import torch


@torch.no_grad()
def encode_images(vision_tower, mm_in_projector, images, **tower_kwargs):
    """image_features, _, _ = vision_tower(images); image_features = mm_in_projector(image_features).

    Returns a RaggedTokens: `feats[i]` is image i's (L_i, hidden) token matrix, which is
    what prepare_inputs_labels_for_multimodal indexes per image (setokim_arch.py:265-293)."""
    image_features, _, _ = vision_tower(images, **tower_kwargs)            # setokim_arch.py:207
    return mm_in_projector(image_features)                                  # :210 (4-D flatten branch :208-209 never taken)


class SetokimVisionMixin:
    """Methods a LLaVA-style meta-model expects (setokim_arch.py:176-211)."""

    def get_vision_tower(self):
        return self.vision_tower

    def get_input_projector(self):
        return self.mm_in_projector

    def encode_images(self, images, **kw):
        return encode_images(self.get_vision_tower(), self.get_input_projector(), images, **kw)
