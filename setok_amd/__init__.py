"""setok_amd — MI355X-native SeTok `encode_images` hot path (ViT tower -> DPC-kNN dynamic clustering ->
cluster encoders -> variable-length tokens) behind the reference's Python surface."""
from .arch import SetokimVisionMixin, encode_images, splice_multimodal
from . import checkpoint
from .builder import build_vision_generator, build_vision_projector, build_vision_tower
from .clip_encoder import CLIPVisionTower
from .detokenizer import SetokDeTokenizer
from .training import HeadTrainer, head_backward, head_forward_train
from .tokenizer import Block, PositionalEncoding2D, RaggedTokens, SetokTokenizer

__all__ = ["SetokTokenizer", "SetokDeTokenizer", "build_vision_generator", "CLIPVisionTower", "Block", "PositionalEncoding2D", "RaggedTokens",
           "build_vision_tower", "build_vision_projector", "encode_images", "splice_multimodal", "HeadTrainer", "head_forward_train", "head_backward", "SetokimVisionMixin", "checkpoint"]
