"""BERT WordPiece tokenizer, written from the published algorithm (Devlin et al. 2018; google-research/bert
``tokenization.py``), with the slice of the Hugging Face tokenizer interface that the reference data path touches.

The reference builds its tokenizer with ``AutoTokenizer.from_pretrained(model_name_or_path, use_fast=False)``
(MarT/data/data_module.py:187, MarT/data/processor.py:256), i.e. ``transformers==4.19.0``'s ``BertTokenizer`` -- a
third-party, un-vendored dependency (requirements.txt:1).  What the path uses of it, and what is restated here:

  * ``tok(text_a, text_b, truncation="longest_first", max_length=L, padding="longest", add_special_tokens=True)``
    on ONE example (processor.py:736-770) -> ``input_ids / token_type_ids / attention_mask`` lists,
    template ``[CLS] A [SEP] B [SEP]`` with segment ids 0.. / 1..;
  * the literal strings ``[UNK] [PAD] [MASK] [SEP]`` inside the text are kept whole (special tokens are cut out of the
    raw text before normalisation) -- the prompt format depends on it (processor.py:127-166, 734, 760-761);
  * ``add_special_tokens({'additional_special_tokens': [...]})`` appending ids after the base vocabulary, ``len(tok)``,
    ``get_added_vocab()`` (data_module.py:191,216-227), ``tok.pad(features, padding, max_length, pad_to_multiple_of,
    return_tensors)`` (data_module.py:113-119), ``*_token`` / ``*_token_id``, ``batch_decode`` / ``decode``.

Normalisation = clean (drop NUL, U+FFFD and category C*; map whitespace to ' '), space out CJK ideographs, NFD + drop
Mn (accents; only in the uncased configuration), lower-case; pre-tokenisation = whitespace split, every punctuation
character (ASCII punctuation or Unicode category P*) its own token; WordPiece = greedy longest-match-first with the
``##`` continuation prefix, whole word -> [UNK] if any part fails or the word is longer than 100 characters.
``longest_first`` truncation follows the slow tokenizer the reference asks for: one token at a time from the longer
sequence, from the pair on a tie.

No vocabulary ships with this package (no network here); ``from_pretrained(dir)`` reads ``dir/vocab.txt``.
tests/test_data_pipeline_cpu.py checks this file token-for-token against the installed ``transformers``/``tokenizers``
BERT implementation on every MarKG entity and relation text.
"""
from __future__ import annotations

import os
import unicodedata
from typing import Dict, Iterable, List, Optional, Sequence, Tuple, Union

import torch

_CJK = ((0x4E00, 0x9FFF), (0x3400, 0x4DBF), (0x20000, 0x2A6DF), (0x2A700, 0x2B73F), (0x2B740, 0x2B81F),
        (0x2B820, 0x2CEAF), (0xF900, 0xFAFF), (0x2F800, 0x2FA1F))


def _is_whitespace(ch: str) -> bool:
    if ch in " \t\n\r":
        return True
    return ch.isspace() or unicodedata.category(ch) == "Zs"


def _is_control(ch: str) -> bool:
    if ch in "\t\n\r":
        return False
    return unicodedata.category(ch).startswith("C")


def _is_punct(ch: str) -> bool:
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp: int) -> bool:
    return any(lo <= cp <= hi for lo, hi in _CJK)


class BertWordPieceTokenizer:
    model_input_names = ["input_ids", "token_type_ids", "attention_mask"]
    padding_side = "right"
    pad_token_type_id = 0

    def __init__(self, vocab: Union[Dict[str, int], Sequence[str]], do_lower_case: bool = True,
                 unk_token: str = "[UNK]", sep_token: str = "[SEP]", pad_token: str = "[PAD]", cls_token: str = "[CLS]",
                 mask_token: str = "[MASK]", max_input_chars_per_word: int = 100, name_or_path: str = ""):
        if not isinstance(vocab, dict):
            vocab = {w: i for i, w in enumerate(vocab)}
        self.vocab: Dict[str, int] = dict(vocab)
        self.ids_to_tokens: Dict[int, str] = {i: w for w, i in self.vocab.items()}
        self.do_lower_case = do_lower_case
        self.unk_token, self.sep_token, self.pad_token = unk_token, sep_token, pad_token
        self.cls_token, self.mask_token = cls_token, mask_token
        self.max_input_chars_per_word = max_input_chars_per_word
        self.name_or_path = name_or_path
        self.added: Dict[str, int] = {}                       # tokens appended after the base vocabulary
        self._special: List[str] = []
        for t in (unk_token, sep_token, pad_token, cls_token, mask_token):
            if t not in self.vocab:
                raise ValueError(f"special token {t} missing from the vocabulary")
            self._special.append(t)
        self._rebuild_matcher()

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_pretrained(cls, name_or_path: str, use_fast: bool = False, **kw) -> "BertWordPieceTokenizer":
        path = os.path.join(name_or_path, "vocab.txt") if os.path.isdir(name_or_path) else name_or_path
        if not os.path.isfile(path):
            raise FileNotFoundError(f"no WordPiece vocabulary at {path!r}: this package ships none (the BERT vocabulary is "
                                    f"a download); point model_name_or_path at a directory holding vocab.txt")
        with open(path, "r", encoding="utf-8") as f:
            words = [line.rstrip("\n") for line in f]
        while words and words[-1] == "":
            words.pop()
        lower = kw.pop("do_lower_case", "uncased" in os.path.basename(os.path.normpath(name_or_path)) or True)
        return cls(words, do_lower_case=lower, name_or_path=name_or_path, **kw)

    def _rebuild_matcher(self) -> None:
        toks = list(dict.fromkeys(self._special + list(self.added)))
        self._cut = sorted(toks, key=len, reverse=True)       # leftmost-longest matching
        self._first = {t[0] for t in self._cut}

    # ------------------------------------------------------------------ properties
    def __len__(self) -> int:
        return len(self.vocab) + len(self.added)

    @property
    def vocab_size(self) -> int:
        return len(self.vocab)

    def _id(self, tok: str) -> int:
        i = self.vocab.get(tok)
        return self.added[tok] if i is None else i

    unk_token_id = property(lambda self: self._id(self.unk_token))
    sep_token_id = property(lambda self: self._id(self.sep_token))
    pad_token_id = property(lambda self: self._id(self.pad_token))
    cls_token_id = property(lambda self: self._id(self.cls_token))
    mask_token_id = property(lambda self: self._id(self.mask_token))

    def get_vocab(self) -> Dict[str, int]:
        v = dict(self.vocab)
        v.update(self.added)
        return v

    def get_added_vocab(self) -> Dict[str, int]:
        return dict(self.added)

    def add_special_tokens(self, special_tokens_dict: Dict[str, Iterable[str]]) -> int:
        """Append new tokens after the current vocabulary (ids len(self), len(self)+1, ...); returns how many were new.
        (data_module.py:191,216: first the 11 292 entity tokens, then the 192 relation tokens.)"""
        n = 0
        for tok in special_tokens_dict.get("additional_special_tokens", []):
            if tok in self.vocab or tok in self.added:
                continue
            self.added[tok] = len(self)
            self.ids_to_tokens[self.added[tok]] = tok
            n += 1
        self._rebuild_matcher()
        return n

    def add_tokens(self, tokens: Iterable[str], special_tokens: bool = True) -> int:
        return self.add_special_tokens({"additional_special_tokens": list(tokens)})

    # ------------------------------------------------------------------ text -> tokens
    def _normalize(self, text: str) -> str:
        out = []
        for ch in text:
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or _is_control(ch):
                continue
            if _is_whitespace(ch):
                out.append(" ")
            elif _is_cjk(cp):
                out.append(" " + ch + " ")
            else:
                out.append(ch)
        text = "".join(out)
        if self.do_lower_case:
            text = "".join(c for c in unicodedata.normalize("NFD", text) if unicodedata.category(c) != "Mn")
            text = text.lower()
        return text

    @staticmethod
    def _pre_tokenize(text: str) -> List[str]:
        words: List[str] = []
        cur: List[str] = []
        for ch in text:
            if _is_whitespace(ch):
                if cur:
                    words.append("".join(cur)); cur = []
            elif _is_punct(ch):
                if cur:
                    words.append("".join(cur)); cur = []
                words.append(ch)
            else:
                cur.append(ch)
        if cur:
            words.append("".join(cur))
        return words

    def _wordpiece(self, word: str) -> List[str]:
        if len(word) > self.max_input_chars_per_word:
            return [self.unk_token]
        pieces: List[str] = []
        start, n = 0, len(word)
        while start < n:
            end = n
            cur = None
            while start < end:
                sub = word[start:end]
                if start > 0:
                    sub = "##" + sub
                if sub in self.vocab:
                    cur = sub
                    break
                end -= 1
            if cur is None:
                return [self.unk_token]
            pieces.append(cur)
            start = end
        return pieces

    def _split_specials(self, text: str) -> List[Tuple[str, bool]]:
        """Cut special / added tokens out of the RAW text (leftmost, longest first); they bypass normalisation."""
        segs: List[Tuple[str, bool]] = []
        i, last, n = 0, 0, len(text)
        while i < n:
            if text[i] in self._first:
                for t in self._cut:
                    if text.startswith(t, i):
                        if i > last:
                            segs.append((text[last:i], False))
                        segs.append((t, True))
                        i += len(t)
                        last = i
                        break
                else:
                    i += 1
            else:
                i += 1
        if last < n:
            segs.append((text[last:], False))
        return segs

    def tokenize(self, text: str) -> List[str]:
        toks: List[str] = []
        for seg, special in self._split_specials(text):
            if special:
                toks.append(seg)
                continue
            for w in self._pre_tokenize(self._normalize(seg)):
                toks.extend(self._wordpiece(w))
        return toks

    def convert_tokens_to_ids(self, tokens):
        if isinstance(tokens, str):
            return self._lookup(tokens)
        return [self._lookup(t) for t in tokens]

    def _lookup(self, t: str) -> int:
        i = self.vocab.get(t)
        if i is None:
            i = self.added.get(t, self.vocab[self.unk_token])
        return i

    def convert_ids_to_tokens(self, ids):
        if isinstance(ids, int):
            return self.ids_to_tokens.get(ids, self.unk_token)
        return [self.ids_to_tokens.get(int(i), self.unk_token) for i in ids]

    # ------------------------------------------------------------------ encoding
    def encode(self, text: str, text_pair: Optional[str] = None, **kw) -> List[int]:
        return self(text, text_pair, **kw)["input_ids"]

    def __call__(self, text: str, text_pair: Optional[str] = None, truncation: Union[bool, str] = False,
                 max_length: Optional[int] = None, padding: Union[bool, str] = False, add_special_tokens: bool = True,
                 **kw) -> Dict[str, List[int]]:
        """One example (processor.py:736-770; ``padding="longest"`` on one example is a no-op) or a list of texts
        (lit_models/transformer.py:46 ``tokenizer(['[R]'], add_special_tokens=False)``) -> lists of lists, padded to the
        longest when ``padding`` asks for it."""
        if isinstance(text, (list, tuple)):
            pairs = text_pair if text_pair is not None else [None] * len(text)
            encs = [self(t, p, truncation=truncation, max_length=max_length, padding=False if padding in (True, "longest") else padding,
                         add_special_tokens=add_special_tokens) for t, p in zip(text, pairs)]
            if padding in (True, "longest"):
                return self.pad(encs, padding="longest")
            return {k: [e[k] for e in encs] for k in encs[0]} if encs else {k: [] for k in self.model_input_names}
        a = self.convert_tokens_to_ids(self.tokenize(text))
        b = self.convert_tokens_to_ids(self.tokenize(text_pair)) if text_pair is not None else None
        n_special = (3 if b is not None else 2) if add_special_tokens else 0
        if truncation and truncation != "do_not_truncate" and max_length is not None:
            over = len(a) + (len(b) if b is not None else 0) + n_special - max_length
            strat = "longest_first" if truncation is True else truncation
            if over > 0:
                if strat == "longest_first":
                    for _ in range(over):
                        if b is None or len(a) > len(b):
                            a = a[:-1]
                        else:
                            b = b[:-1]
                elif strat == "only_first":
                    a = a[:max(0, len(a) - over)]
                elif strat == "only_second" and b is not None:
                    b = b[:max(0, len(b) - over)]
                else:
                    raise ValueError(f"unknown truncation strategy {truncation!r}")
        if add_special_tokens:
            ids = [self.cls_token_id] + a + [self.sep_token_id]
            tt = [0] * len(ids)
            if b is not None:
                ids += b + [self.sep_token_id]
                tt += [1] * (len(b) + 1)
        else:
            ids = a + (b or [])
            tt = [0] * len(a) + [1] * len(b or [])
        enc = {"input_ids": ids, "token_type_ids": tt, "attention_mask": [1] * len(ids)}
        if padding == "max_length" and max_length is not None and len(ids) < max_length:
            pad = max_length - len(ids)
            enc["input_ids"] = ids + [self.pad_token_id] * pad
            enc["token_type_ids"] = tt + [self.pad_token_type_id] * pad
            enc["attention_mask"] = enc["attention_mask"] + [0] * pad
        return enc

    def pad(self, features: Sequence[Dict[str, List[int]]], padding: Union[bool, str] = True,
            max_length: Optional[int] = None, pad_to_multiple_of: Optional[int] = None,
            return_tensors: Optional[str] = None):
        """Right-pad a list of encoded examples to a common length (data_module.py:113-119): ``longest`` -> longest of the
        batch, ``max_length`` -> ``max_length``; then up to a multiple of ``pad_to_multiple_of``."""
        keys = [k for k in self.model_input_names if k in features[0]]
        other = [k for k in features[0] if k not in keys]
        if padding is True:
            padding = "longest"
        if padding == "longest":
            target = max(len(f["input_ids"]) for f in features)
        elif padding == "max_length":
            target = max_length
        elif padding in (False, "do_not_pad"):
            target = None
        else:
            raise ValueError(f"unknown padding strategy {padding!r}")
        if target is not None and pad_to_multiple_of and target % pad_to_multiple_of:
            target = (target // pad_to_multiple_of + 1) * pad_to_multiple_of
        fill = {"input_ids": self.pad_token_id, "token_type_ids": self.pad_token_type_id, "attention_mask": 0}
        out: Dict[str, list] = {k: [] for k in keys + other}
        for f in features:
            n = len(f["input_ids"])
            for k in keys:
                row = list(f[k])
                if target is not None and n < target:
                    row = row + [fill[k]] * (target - n)
                out[k].append(row)
            for k in other:
                out[k].append(f[k])
        if return_tensors == "pt":
            for k in keys:
                out[k] = torch.tensor(out[k], dtype=torch.long)
        elif return_tensors not in (None, "np"):
            raise ValueError("return_tensors must be 'pt' or None")
        return out

    # ------------------------------------------------------------------ decoding
    def decode(self, ids, skip_special_tokens: bool = False) -> str:
        toks = self.convert_ids_to_tokens([int(i) for i in ids])
        if skip_special_tokens:
            sp = set(self._special) | set(self.added)
            toks = [t for t in toks if t not in sp]
        text = " ".join(toks).replace(" ##", "")
        return text

    def batch_decode(self, seqs, skip_special_tokens: bool = False) -> List[str]:
        return [self.decode(s, skip_special_tokens) for s in seqs]

    def save_vocabulary(self, directory: str) -> str:
        path = os.path.join(directory, "vocab.txt")
        with open(path, "w", encoding="utf-8") as f:
            for w, _ in sorted(self.vocab.items(), key=lambda kv: kv[1]):
                f.write(w + "\n")
        return path
