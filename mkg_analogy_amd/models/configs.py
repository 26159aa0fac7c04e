"""Minimal stand-ins for transformers' BertConfig / CLIPVisionConfig (MarT/main.py:79-80).

The model accepts either these or the real HF config objects: only attribute access is used.
Defaults = bert-base-uncased and clip-vit-base-patch32 (patch_size=16 gives the 196-patch BASELINE config).
"""
from dataclasses import dataclass


@dataclass
class TextConfig:
    vocab_size: int = 30522
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    hidden_act: str = "gelu"
    hidden_dropout_prob: float = 0.1
    attention_probs_dropout_prob: float = 0.1
    max_position_embeddings: int = 512
    type_vocab_size: int = 2
    initializer_range: float = 0.02
    layer_norm_eps: float = 1e-12
    pad_token_id: int = 0
    torchscript: bool = False
    chunk_size_feed_forward: int = 0
    add_cross_attention: bool = False


@dataclass
class VisionConfig:
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    image_size: int = 224
    patch_size: int = 32
    hidden_act: str = "quick_gelu"
    layer_norm_eps: float = 1e-5
    attention_dropout: float = 0.0
    device: str = "cpu"
