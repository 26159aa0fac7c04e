"""MARS / MarKG prompt and feature pipeline behind the reference's ``data.processor`` names.

Mirrors MarT/data/processor.py: ``KGProcessor`` (same public methods), ``get_dataset(args, processor, mode)`` (same
cache file naming, processor.py:53-56), ``KGCDataset``.  What is reproduced, with the reference line it follows:

  * id tables: entity id = line order of entity2text(long).txt (:610-616, long file preferred :509-510); relation id =
    line order of relation2text.txt, EXCEPT that analogy relations are re-indexed 0.. in file order (:639-643 -- the
    ``rel_label`` quirk); ``analogy_ent2id`` = rank among analogy entities in entity-file order (:629-633);
  * fine-tune prompts per ``mode`` 0/1/2 (:155-217): six segments joined by ``[SEP]`` into the two tokenizer inputs
    ``[UNK] h [SEP] [PAD] [SEP] [UNK] t`` / ``[UNK] q [SEP] [PAD] [SEP] [MASK]`` with text or image standing in for
    each entity, and which entities feed the two image slots;
  * pre-train prompts (:98-149): per triple one ``random.random()`` draw (module RNG seeded 1, :10) picks
    (text,text) <=0.4, (image,text) <0.7, (image,image); two examples, pre_type 1 (tail prediction) and 2 (relation);
  * id plumbing after tokenisation (:269-319): [UNK] placeholders -> ``len(tokenizer) + entity id`` (a FRESH tokenizer,
    so 30522 + id, :256), [PAD] placeholders -> the relation token; fine-tune records sep_idx / rel_idx / q_head_idx /
    a_head_idx.  Pre-train examples never carry rel_id / tail_id (they stay -1, :336-337): only the first [UNK] is
    replaced and the first [PAD] becomes ``30522 + E - 1`` -- kept as is.

The structure is not the reference's (no globals, no process pool, no example classes): tables are loaded once into
``MarsTables``, prompts are plain tuples, and the result can additionally be packed into flat int arrays
(``KGCDataset.packed()``) for the device-side batch assembly in data_module.py.
"""
from __future__ import annotations

import json
import os
import pickle
import random
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .tokenization import BertWordPieceTokenizer

_rng = random.Random(1)        # processor.py:10 -- one stream for the whole process, shared by train/dev/test builds


def reseed(seed: int = 1) -> None:
    _rng.seed(seed)


def _read_pairs(path: str) -> Dict[str, str]:
    """key<TAB>value per line; the value loses its last character (the newline) exactly like processor.py:490-491."""
    out: Dict[str, str] = {}
    with open(path, "r", encoding="utf-8") as f:
        for line in f.readlines():
            key, value = line.split("\t")
            out[key] = value[:-1]
    return out


def _read_list(path: str) -> List[str]:
    with open(path, "r") as f:
        return [line.strip().replace("\n", "") for line in f.readlines()]


@dataclass
class MarsTables:
    ent2text: Dict[str, str]
    ent2id: Dict[str, int]
    rel2text: Dict[str, str]
    rel2id: Dict[str, int]
    analogy_ent2id: Dict[str, int]

    @classmethod
    def load(cls, entity_path: str, pretrain_path: str, data_dir: str) -> "MarsTables":
        ent2text = _read_pairs(entity_path)
        ent2id = {e: i for i, e in enumerate(ent2text)}
        rel2text = _read_pairs(os.path.join(pretrain_path, "relation2text.txt"))
        rel2id = {r: i for i, r in enumerate(rel2text)}
        analogy_entities = set(_read_list(os.path.join(data_dir, "analogy_entities.txt")))
        analogy_ent2id, i = {}, 0
        for e in ent2text:
            if e in analogy_entities:
                analogy_ent2id[e] = i
                i += 1
        analogy_relations = set(_read_list(os.path.join(data_dir, "analogy_relations.txt")))
        i = 0
        for r in rel2id:                                        # :639-643 analogy relations get ids 0..n-1 in file order
            if r in analogy_relations:
                rel2id[r] = i
                i += 1
        return cls(ent2text, ent2id, rel2text, rel2id, analogy_ent2id)


# a prompt = (segments of input A, segments of input B or None, fields carried into the feature dict)
Prompt = Tuple[List[str], Optional[List[str]], Dict[str, object]]


def analogy_prompt(line: Dict[str, object], t: MarsTables) -> Prompt:
    """One MARS line {"example": [h, t], "question": q, "answer": a, "relation": r, "mode": m} (processor.py:151-217)."""
    head, tail = line["example"]
    question, answer, rel, mode = line["question"], line["answer"], line["relation"], line["mode"]
    if mode == 0:        # (T, T) -> (I, ?)
        a = ["[UNK] " + t.ent2text[head], "[PAD]", "[UNK] " + t.ent2text[tail]]
        b = ["[UNK] ", "[PAD]", "[MASK]"]
        head_ent, tail_ent = question, None
    elif mode == 1:      # (I, I) -> (T, ?)
        a = ["[UNK] ", "[PAD]", "[UNK] "]
        b = ["[UNK] " + t.ent2text[question], "[PAD]", "[MASK]"]
        head_ent, tail_ent = head, tail
    elif mode == 2:      # (I, T) -> (I, ?)
        a = ["[UNK] ", "[PAD]", "[UNK] " + t.ent2text[tail]]
        b = ["[UNK] ", "[PAD]", "[MASK]"]
        head_ent, tail_ent = head, question
    else:
        raise ValueError(f"unknown MARS mode {mode!r}")
    fields = dict(label=t.analogy_ent2id[answer], rel_label=t.rel2id[rel], q_head_id=t.ent2id[head],
                  q_tail_id=t.ent2id[tail], a_head_id=t.ent2id[question], head_ent=head_ent, tail_ent=tail_ent)
    return a, b, fields


def pretrain_prompts(triple: Sequence[str], t: MarsTables, rng: random.Random) -> List[Prompt]:
    """One MarKG triple -> (head, rel, [MASK]) and (head, [MASK], tail) (processor.py:98-149)."""
    head, rel, tail = triple
    rnd = rng.random()
    if rnd <= 0.4:
        head_text, tail_text, head_ent, tail_ent = t.ent2text[head], t.ent2text[tail], None, None
    elif rnd < 0.7:
        head_text, tail_text, head_ent, tail_ent = "", t.ent2text[tail], head, None
    else:
        head_text, tail_text, head_ent, tail_ent = "", "", head, tail
    common = dict(head_id=t.ent2id[head], rel_id=-1, tail_id=-1, head_ent=head_ent)
    p1 = (["[UNK] " + head_text, "[PAD] " + t.rel2text[rel], "[MASK]"], None,
          dict(common, label=t.ent2id[tail], tail_ent=None, pre_type=1))
    p2 = (["[UNK] " + head_text, "[MASK]", "[UNK] " + tail_text], None,
          dict(common, label=t.rel2id[rel], tail_ent=tail_ent, pre_type=2))
    return [p1, p2]


def encode_prompt(tok, prompt: Prompt, max_seq_length: int) -> Dict[str, object]:
    a, b, fields = prompt
    text_a = tok.sep_token.join(a)
    text_b = tok.sep_token.join(b) if b is not None else None
    enc = tok(text_a, text_b, truncation="longest_first", max_length=max_seq_length, padding="longest", add_special_tokens=True)
    if tok.mask_token_id not in enc["input_ids"]:
        raise AssertionError("mask token must in input")        # processor.py:785
    feat: Dict[str, object] = {"input_ids": list(enc["input_ids"]), "attention_mask": list(enc["attention_mask"]),
                               "token_type_ids": list(enc["token_type_ids"])}
    feat.update(fields)
    return feat


def plumb_finetune(features: List[Dict[str, object]], tok, num_entities: int, num_relations: int) -> None:
    """processor.py:293-319.  The position variables live across examples like the reference's loop variables."""
    base = len(tok)
    unk, sep, pad = tok.unk_token_id, tok.sep_token_id, tok.pad_token_id
    r_token = base + num_entities + num_relations
    q_head_idx = a_head_idx = None
    for f in features:
        ent = [f.pop("q_head_id"), f.pop("q_tail_id"), f.pop("a_head_id")]
        ids = f["input_ids"]
        count, sep_idx = 0, []
        for i, tid in enumerate(list(ids)):
            if count < 3 and tid == unk:
                ids[i] = ent[count] + base
                if count == 0:
                    q_head_idx = i
                elif count == 2:
                    a_head_idx = i
                count += 1
            if tid == sep:
                sep_idx.append(i)
        if q_head_idx is None or a_head_idx is None:
            raise NameError("fewer than three [UNK] placeholders in the first example")   # what the reference would hit
        f["sep_idx"], f["q_head_idx"], f["a_head_idx"] = sep_idx, q_head_idx, a_head_idx
        rel_idx = []
        for i, tid in enumerate(list(ids)):
            if tid == pad:
                ids[i] = r_token
                rel_idx.append(i)
        f["rel_idx"] = rel_idx


def plumb_pretrain(features: List[Dict[str, object]], tok, num_entities: int) -> None:
    """processor.py:269-291."""
    base = len(tok)
    unk, pad = tok.unk_token_id, tok.pad_token_id
    for f in features:
        head_id, rel_id, tail_id = f.pop("head_id"), f.pop("rel_id"), f.pop("tail_id")
        ids = f["input_ids"]
        if head_id != -1 and tail_id != -1:
            ent, count = [head_id, tail_id], 0
            for i, tid in enumerate(list(ids)):
                if tid == unk and count < 2:
                    ids[i] = ent[count] + base
                    count += 1
        else:
            ent_id = head_id if head_id != -1 else tail_id
            for i, tid in enumerate(ids):
                if tid == unk:
                    ids[i] = ent_id + base
                    break
        for i, tid in enumerate(ids):
            if tid == pad:
                ids[i] = rel_id + base + num_entities
                break


class KGCDataset:
    """List of feature dicts (what the reference pickles, processor.py:684-692) + a packed view for fast batching."""

    def __init__(self, features: List[Dict[str, object]]):
        self.features = features
        self._packed = None

    def __getitem__(self, index):
        return self.features[index]

    def __len__(self):
        return len(self.features)

    def packed(self) -> Dict[str, np.ndarray]:
        """Flat int32 token arrays + offsets and [N,...] index arrays: batches become numpy slices instead of dict walks."""
        if self._packed is None:
            f = self.features
            lens = np.array([len(x["input_ids"]) for x in f], dtype=np.int64)
            off = np.zeros(len(f) + 1, dtype=np.int64)
            np.cumsum(lens, out=off[1:])
            p = {"offsets": off,
                 "input_ids": np.fromiter((t for x in f for t in x["input_ids"]), dtype=np.int32, count=int(off[-1])),
                 "token_type_ids": np.fromiter((t for x in f for t in x["token_type_ids"]), dtype=np.int8, count=int(off[-1])),
                 "label": np.array([x["label"] for x in f], dtype=np.int64)}
            for k in ("rel_label", "pre_type", "q_head_idx", "a_head_idx"):
                if f and k in f[0]:
                    p[k] = np.array([x[k] for x in f], dtype=np.int64)
            for k in ("sep_idx", "rel_idx"):
                if f and k in f[0]:
                    p[k] = np.array([x[k] for x in f], dtype=np.int64)       # raises if ragged, like torch.tensor would
            self._packed = p
        return self._packed


class _RefUnpickler(pickle.Unpickler):
    """Caches written by the reference name the class ``data.processor.KGCDataset``."""

    def find_class(self, module, name):
        if name == "KGCDataset" and module.endswith("processor"):
            return KGCDataset
        return super().find_class(module, name)


class KGProcessor:
    """Processor for knowledge graph data set (processor.py:503-681, same method names)."""

    def __init__(self, tokenizer, args):
        self.labels = set()
        self.tokenizer = tokenizer
        self.args = args
        long_path = os.path.join(args.pretrain_path, "entity2textlong.txt")
        self.entity_path = long_path if os.path.exists(long_path) else os.path.join(args.pretrain_path, "entity2text.txt")
        self._tables: Optional[MarsTables] = None

    # ---- tables
    def tables(self, data_dir: str) -> MarsTables:
        if self._tables is None:
            self._tables = MarsTables.load(self.entity_path, self.args.pretrain_path, data_dir)
        return self._tables

    def _entity_keys(self) -> List[str]:
        with open(self.entity_path, "r") as f:
            return [line.strip().split("\t")[0] for line in f.readlines()]

    def _relation_keys(self) -> List[str]:
        with open(os.path.join(self.args.pretrain_path, "relation2text.txt"), "r") as f:
            return [line.strip().split("\t")[0] for line in f.readlines()]

    def get_entities(self, data_dir):
        keys = self._entity_keys()
        return list({e: f"[ENTITY_{i}]" for i, e in enumerate(keys)}.values())

    def get_relations(self, data_dir):
        keys = self._relation_keys()
        return list({r: f"[RELATION_{i}]" for i, r in enumerate(keys)}.values())

    def get_analogy_entities(self, data_dir):
        chosen = set(_read_list(os.path.join(data_dir, "analogy_entities.txt")))
        return list({e: f"[ENTITY_{i}]" for i, e in enumerate(self._entity_keys()) if e in chosen}.values())

    def get_analogy_relations(self, data_dir):
        chosen = set(_read_list(os.path.join(data_dir, "analogy_relations.txt")))
        return list({r: f"[RELATION_{i}]" for i, r in enumerate(self._relation_keys()) if r in chosen}.values())

    def get_labels(self, data_dir):
        with open(os.path.join(self.args.pretrain_path, "relation2text.txt"), "r") as f:
            return [line.strip().split("\t")[-1] for line in f.readlines()]

    # ---- prompts
    def _lines(self, data_dir: str, split: str):
        if self.args.pretrain:
            out = []
            with open(os.path.join(self.args.pretrain_path, "wiki_tuple_ids.txt"), "r", encoding="utf-8") as f:
                for line in f.readlines():
                    head, rel, tail = line.split("\t")
                    out.append((head, rel, tail.replace("\n", "")))
            return out
        with open(os.path.join(data_dir, f"{split}.json"), "r", encoding="utf-8") as f:
            return [json.loads(line) for line in f.readlines()]

    def _create_examples(self, lines, set_type, data_dir, args) -> List[Prompt]:
        t = self.tables(data_dir)
        if args.pretrain:
            kept = [l for l in lines if l[0] in t.ent2text and l[2] in t.ent2text and l[1] in t.rel2text]   # :653-656
            return [p for l in kept for p in pretrain_prompts(l, t, _rng)]
        return [analogy_prompt(l, t) for l in lines]

    def get_train_examples(self, data_dir):
        return self._create_examples(self._lines(data_dir, "train"), "train", data_dir, self.args)

    def get_dev_examples(self, data_dir):
        return self._create_examples(self._lines(data_dir, "dev"), "dev", data_dir, self.args)

    def get_test_examples(self, data_dir, chunk=""):
        return self._create_examples(self._lines(data_dir, "test"), "test", data_dir, self.args)


def build_features(args, processor: KGProcessor, mode: str, tokenizer=None) -> List[Dict[str, object]]:
    """Prompts -> tokens -> id plumbing; ``tokenizer`` defaults to a FRESH one (no entity tokens added), as :256."""
    assert mode in ["train", "dev", "test"], "mode must be in train dev test!"
    prompts = {"train": processor.get_train_examples, "dev": processor.get_dev_examples,
               "test": processor.get_test_examples}[mode](args.data_dir)
    tok = tokenizer if tokenizer is not None else BertWordPieceTokenizer.from_pretrained(args.model_name_or_path, use_fast=False)
    features = [encode_prompt(tok, p, args.max_seq_length) for p in prompts]
    num_entities = len(processor.get_entities(args.data_dir))
    num_relations = len(processor.get_relations(args.data_dir))
    if args.pretrain:
        plumb_pretrain(features, tok, num_entities)
    else:
        plumb_finetune(features, tok, num_entities, num_relations)
    return features


def cache_path(args, mode: str) -> str:
    model_name = args.model_name_or_path.split("/")[-1]
    return os.path.join(args.data_dir, f"cached_{mode}_features{model_name}_pretrain{args.pretrain}.pkl")


def get_dataset(args, processor: KGProcessor, mode: str, tokenizer=None) -> KGCDataset:
    """Same contract as processor.py:243-321 incl. the cache file (name, location, ``overwrite_cache``)."""
    path = cache_path(args, mode)
    if not getattr(args, "overwrite_cache", False) and os.path.exists(path):
        with open(path, "rb") as f:
            return _RefUnpickler(f).load()
    ds = KGCDataset(build_features(args, processor, mode, tokenizer))
    try:
        with open(path, "wb") as f:
            pickle.dump(ds, f)
    except OSError:
        pass                                                    # read-only data directory: keep going without a cache
    return ds
