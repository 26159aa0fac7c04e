from .model import FlavaKGC, MKGformerKGC
from .modeling_unimo import UnimoForMaskedLM
from .modeling_flava import FlavaForMaskedLM, flava_config
from .configs import TextConfig, VisionConfig

__all__ = ["MKGformerKGC", "FlavaKGC", "UnimoForMaskedLM", "FlavaForMaskedLM", "flava_config", "TextConfig", "VisionConfig"]
