from .transformer import TransformerLitModel

__all__ = ["TransformerLitModel"]
