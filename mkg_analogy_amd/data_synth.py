"""Seeded synthetic MARS-shaped batches; the real data pipeline (MarT/data) is out of scope for this path.

Prompt layout produced by the reference's processor (MarT/data/processor.py:150-216):
  [CLS] E_h d.. [SEP] [R] [SEP] E_t d.. [SEP] E_q d.. [SEP] [R] [SEP] [MASK] [SEP] [PAD]..
ids: BERT specials 0/100/101/102/103, entity i -> 30522+i, relation j -> 30522+11292+j, [R] -> 42006.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

BASE_VOCAB, N_ENT, N_REL = 30522, 11292, 192
VOCAB = BASE_VOCAB + N_ENT + N_REL + 1
R_TOKEN = VOCAB - 1
N_ANALOGY = 2063
CLS, SEP, MASK = 101, 102, 103


class FakeTokenizer:
    """What TransformerLitModel needs from a tokenizer when no WordPiece vocabulary is available offline."""
    mask_token_id = MASK
    pad_token_id = 0

    def __init__(self, n: int = VOCAB - 1):
        self.n = n
        self.extra: Dict[str, int] = {}

    def __len__(self):
        return self.n + len(self.extra)

    def add_special_tokens(self, d):
        k = 0
        for t in d.get("additional_special_tokens", []):
            if t not in self.extra:
                self.extra[t] = self.n + len(self.extra)
                k += 1
        return k

    def __call__(self, texts, add_special_tokens=False):
        return {"input_ids": [[self.extra[t]] for t in texts]}

    def batch_decode(self, ids, **k):
        return [" ".join(str(int(t)) for t in row) for row in ids]


def data_config(seed: int = 1234) -> Dict[str, object]:
    rng = np.random.default_rng(seed)
    ent = np.sort(rng.choice(np.arange(BASE_VOCAB, BASE_VOCAB + N_ENT), size=N_ANALOGY, replace=False))
    rel = np.sort(rng.choice(np.arange(BASE_VOCAB + N_ENT, BASE_VOCAB + N_ENT + N_REL), size=27, replace=False))
    return dict(entity_id_st=BASE_VOCAB, entity_id_ed=BASE_VOCAB + N_ENT, relation_id_st=BASE_VOCAB + N_ENT,
                relation_id_ed=BASE_VOCAB + N_ENT + N_REL, analogy_entity_ids=ent.tolist(), analogy_relation_ids=rel.tolist())


def make_batch(B: int, L: int, image_size: int = 224, seed: int = 1234, device="cpu", pretrain: bool = False,
               n_labels: int = N_ANALOGY) -> Dict[str, torch.Tensor]:
    rng = np.random.default_rng(seed)
    ids = np.zeros((B, L), np.int64)
    am = np.zeros((B, L), np.int64)
    tt = np.zeros((B, L), np.int64)
    sep = np.zeros((B, 6), np.int64)
    rel = np.zeros((B, 2), np.int64)
    qh, ah = np.zeros(B, np.int64), np.zeros(B, np.int64)
    for b in range(B):
        real = int(rng.integers(min(40, L - 4), L + 1))
        free = max(real - 13, 0)
        c = np.sort(rng.integers(0, free + 1, size=2))
        d = [int(c[0]), int(c[1] - c[0]), int(free - c[1])]
        e = rng.integers(BASE_VOCAB, BASE_VOCAB + N_ENT, size=3)
        desc = lambda n: rng.integers(1000, BASE_VOCAB, size=n).tolist()
        seq = [CLS, int(e[0])] + desc(d[0]) + [SEP, R_TOKEN, SEP, int(e[1])] + desc(d[1]) + [SEP, int(e[2])] + desc(d[2]) + [SEP, R_TOKEN, SEP, MASK, SEP]
        n = len(seq)
        ids[b, :n], am[b, :n] = seq, 1
        sp = [i for i, t in enumerate(seq) if t == SEP]
        sep[b] = sp
        tt[b, sp[2] + 1:n] = 1
        rel[b] = [i for i, t in enumerate(seq) if t == R_TOKEN]
        qh[b], ah[b] = 1, sp[2] + 1
    g = torch.Generator().manual_seed(seed)
    pix = torch.randn((B, 2, 3, image_size, image_size), generator=g)
    drop = torch.from_numpy(rng.random(B) < 0.4)
    pix[drop, 1] = 0.0
    out = dict(input_ids=ids, attention_mask=am, token_type_ids=tt, sep_idx=sep, rel_idx=rel, q_head_idx=qh, a_head_idx=ah,
               label=rng.integers(0, n_labels, size=B), rel_label=rng.integers(0, 27, size=B))
    out = {k: torch.from_numpy(np.asarray(v)) for k, v in out.items()}
    out["pixel_values"] = pix
    if pretrain:
        out.pop("sep_idx")
        pt = rng.integers(1, 3, size=B)
        out["pre_type"] = torch.from_numpy(pt)
        out["label"] = torch.from_numpy(np.where(pt == 2, rng.integers(0, N_REL, size=B), rng.integers(0, N_ENT, size=B)))
    return {k: v.to(device) for k, v in out.items()}


def synthetic_wordpiece_vocab(texts, size: int = 30522):
    """A deterministic stand-in for bert-base-uncased's vocab.txt (not downloadable here): same size and the same ids for
    the specials ([PAD]=0, [UNK]=100, [CLS]=101, [SEP]=102, [MASK]=103), body built from ``texts`` -- every character and
    ``##``character (so nothing degenerates to [UNK]), words seen >= 3 times whole, rarer words as a 3-4 letter stem plus
    ``##`` suffix pieces -- so that WordPiece really splits.  Only used by tests / fixture generation."""
    from collections import Counter
    from .data.tokenization import BertWordPieceTokenizer
    base = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"] + \
           [f"[unused{i}]" for i in range(99, 994)]
    probe = BertWordPieceTokenizer(base)
    words = Counter()
    for t in texts:
        words.update(probe._pre_tokenize(probe._normalize(t)))
    chars = sorted({c for w in words for c in w})
    vocab = list(base)
    seen = set(vocab)

    def add(tok):
        if tok not in seen and len(vocab) < size:
            seen.add(tok); vocab.append(tok)
    for c in chars:
        add(c)
    for c in chars:
        add("##" + c)
    for w, n in sorted(words.items(), key=lambda kv: (-kv[1], kv[0])):
        if n >= 3:
            add(w)
    for w, n in sorted(words.items(), key=lambda kv: (-kv[1], kv[0])):
        if n < 3 and len(w) > 4:
            k = 3 + (len(w) & 1)
            add(w[:k]); add("##" + w[k:k + 3])
            if len(w) > k + 3:
                add("##" + w[k + 3:])
    i = 994
    while len(vocab) < size:
        vocab.append(f"[unused{i}]"); i += 1
    return vocab


def seeded_weights(named_shapes, seed: int = 0, conditioned: bool = False) -> Dict[str, torch.Tensor]:
    """The seeded random-init weight set of the reference goldens G7 / G8: numpy PCG64 draws in ``named_parameters()`` order --
    N(0, 0.02) matrices / embeddings / biases, LayerNorm weights 1 + 0.05 N(0,1), adaptive_weight = (0.25, 0.5).  ``conditioned``: the text value
    projections of layers 8-11 scaled by 0.05, which takes the UNSCALED fusion softmax of those layers (modeling_unimo.py:405-410) out of its
    one-hot, chaotic regime.  Every side (golden generator, CPU oracle, this package) regenerates the same tensors from the seed;
    tests/test_host_logic_cpu.py holds this function to the oracle's bit for bit."""
    rng = np.random.default_rng(seed)
    out: Dict[str, torch.Tensor] = {}
    for name, shape in named_shapes:
        shape = tuple(shape)
        if name.endswith("adaptive_weight.0"):
            v = np.full(shape, 0.25, np.float32)
        elif name.endswith("adaptive_weight.1"):
            v = np.full(shape, 0.5, np.float32)
        elif any(t in name for t in ("LayerNorm.weight", "layer_norm1.weight", "layer_norm2.weight", "layrnorm.weight", "layernorm.weight")):
            v = (1.0 + 0.05 * rng.standard_normal(shape)).astype(np.float32)
        else:
            v = (0.02 * rng.standard_normal(shape)).astype(np.float32)
        out[name] = torch.from_numpy(v)
    if conditioned:
        for l in range(8, 12):
            for k in ("weight", "bias"):
                n = f"unimo.encoder.text_layer.{l}.attention.self.value.{k}"
                out[n] = out[n] * 0.05
    return out


def load_seeded_weights(model, lit, seed: int = 0, conditioned: bool = False) -> None:
    """Put ``seeded_weights`` into a finalized MKGformer model whose vocabulary already carries the [R] row (bench.py): the set is drawn for the
    vocabulary WITHOUT [R] (as the goldens' generator loads it before ``_init_relation_word``); the [R] row is then rebuilt from the analogy
    relation rows (lit_models/transformer.py:41-54) and its decoder-bias entry is zero (the zero-padded resize, modeling_unimo.py:915-930)."""
    named = [(n, p.shape) for n, p in model.named_parameters()]
    v0 = VOCAB - 1
    shapes = [(n, ((v0,) + tuple(s[1:]) if n in ("unimo.text_embeddings.word_embeddings.weight", "cls.predictions.bias") else tuple(s))) for n, s in named]
    sd = seeded_weights(shapes, seed, conditioned)
    with torch.no_grad():
        for n, p in model.named_parameters():
            src = sd[n].to(p.device)
            if src.shape[0] != p.shape[0]:
                p[:src.shape[0]].copy_(src)
                p[src.shape[0]:].zero_()
            else:
                p.copy_(src)
    lit._init_relation_word()                       # rebuilds the [R] row and refreshes the bf16 / fp16 / transposed shadows
