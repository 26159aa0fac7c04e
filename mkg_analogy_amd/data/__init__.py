from .tokenization import BertWordPieceTokenizer  # noqa: F401
