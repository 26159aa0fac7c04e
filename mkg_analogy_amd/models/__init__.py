from .model import MKGformerKGC
from .modeling_unimo import UnimoForMaskedLM
from .configs import TextConfig, VisionConfig

__all__ = ["MKGformerKGC", "UnimoForMaskedLM", "TextConfig", "VisionConfig"]
