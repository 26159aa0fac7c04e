"""MARS / MarKG prompt + feature pipeline (BASELINE configs[0] plumbing) against G6, the outputs of
the unmodified reference ``data`` package on the fixtures in tests/golden/mars/ (oracle/gen_goldens_data.py).
Integer work: everything is compared bit-exact."""
import argparse
import copy
import os
import shutil

import numpy as np
import pytest
import torch

from mkg_analogy_amd.data import BertWordPieceTokenizer
from mkg_analogy_amd.data import processor as P
from mkg_analogy_amd.data.data_module import KGC
from mkg_analogy_amd.data_synth import synthetic_wordpiece_vocab

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = os.path.join(HERE, "golden", "mars")
G6 = np.load(os.path.join(HERE, "golden", "g6_mars_plumbing.npz"))


def _texts():
    out = []
    for fn in ("entity2text.txt", "relation2text.txt"):
        with open(os.path.join(FIX, fn), encoding="utf-8") as f:
            out += [line.split("\t", 1)[1][:-1] for line in f]
    return out


@pytest.fixture(scope="module")
def dirs(tmp_path_factory):
    tmp = tmp_path_factory.mktemp("mars")
    data_dir, pre = tmp / "MARS", tmp / "MarKG"
    data_dir.mkdir(); pre.mkdir()
    for fn in ("dev.json", "train.json", "test.json", "analogy_entities.txt", "analogy_relations.txt"):
        shutil.copy(os.path.join(FIX, fn), data_dir)
    for fn in ("entity2text.txt", "relation2text.txt", "wiki_tuple_ids.txt"):
        shutil.copy(os.path.join(FIX, fn), pre)
    return str(data_dir), str(pre)


def _args(dirs, pretrain=0, L=64):
    return argparse.Namespace(model_name_or_path="bert-base-uncased", data_dir=dirs[0], pretrain_path=dirs[1], pretrain=pretrain,
                              max_seq_length=L, overwrite_cache=True, precision=32, model_class="MKGformerKGC",
                              batch_size=8, eval_batch_size=8, num_workers=0)


def _vis():
    n = sum(1 for _ in open(os.path.join(FIX, "entity2text.txt")))
    return torch.randn(n, 3, 4, 4, generator=torch.Generator().manual_seed(int(G6["visual_seed"])))


def test_synthetic_vocab_is_the_fixture_vocab():
    assert synthetic_wordpiece_vocab(_texts()) == G6["vocab"].tolist()


def test_wordpiece_matches_installed_bert_tokenizer():
    """Token-for-token against transformers' BERT tokenizer (tokenizers backend) on every fixture text, wrapped in the
    prompt format (literal special tokens inside the text, pair input, truncation)."""
    tr = pytest.importorskip("transformers")
    vocab = G6["vocab"].tolist()
    hf = tr.BertTokenizer(vocab={w: i for i, w in enumerate(vocab)})
    mine = BertWordPieceTokenizer(vocab)
    extra = ["Café Ünïcode — naïve “quotes” 東京タワー x­soft​zero\ttab", "a" * 120 + " ok", "", "  ", "[MASK][SEP]x[PAD]",
             "ǅ İstanbul ΣΑΣ ß ﬁ 𝒳 ́lead", "3.14%$^~`|<=>"]
    for t in _texts() + extra:
        a = mine("[UNK] " + t + "[SEP][PAD]", "[UNK] " + t[:15] + "[SEP][MASK]", truncation="longest_first", max_length=200)
        b = hf("[UNK] " + t + "[SEP][PAD]", "[UNK] " + t[:15] + "[SEP][MASK]", truncation="longest_first", max_length=200)
        assert a["input_ids"] == b["input_ids"], t
        assert a["token_type_ids"] == b["token_type_ids"] and a["attention_mask"] == b["attention_mask"]


def test_tokenizer_interface():
    tok = BertWordPieceTokenizer(G6["vocab"].tolist())
    assert (len(tok), tok.pad_token_id, tok.unk_token_id, tok.cls_token_id, tok.sep_token_id, tok.mask_token_id) == \
           (30522, 0, 100, 101, 102, 103)
    assert tok.add_special_tokens({"additional_special_tokens": ["[ENTITY_0]", "[ENTITY_1]"]}) == 2
    assert tok.get_added_vocab() == {"[ENTITY_0]": 30522, "[ENTITY_1]": 30523} and len(tok) == 30524
    assert tok("x [ENTITY_1] y")["input_ids"][2] == 30523
    assert tok(["[ENTITY_1]", "[ENTITY_0] [ENTITY_1]"], add_special_tokens=False)["input_ids"] == [[30523], [30522, 30523]]
    # slow-tokenizer longest_first: one token at a time from the longer input, from the pair on a tie
    e = tok("a b c d e f", "a b c d e f", truncation="longest_first", max_length=10)
    assert e["token_type_ids"] == [0] * 6 + [1] * 4                      # 7 of 12 tokens survive: 4 of A, 3 of B
    p = tok.pad([{"input_ids": [1, 2, 3], "attention_mask": [1, 1, 1], "token_type_ids": [0, 1, 1]},
                 {"input_ids": [4], "attention_mask": [1], "token_type_ids": [0]}], padding="longest", pad_to_multiple_of=8,
                return_tensors="pt")
    assert p["input_ids"].tolist() == [[1, 2, 3, 0, 0, 0, 0, 0], [4, 0, 0, 0, 0, 0, 0, 0]]
    assert p["attention_mask"].sum().item() == 4
    with pytest.raises(FileNotFoundError):
        BertWordPieceTokenizer.from_pretrained("/nonexistent/bert-base-uncased")


def _check_split(ds, prefix):
    pk = ds.packed()
    assert np.array_equal(pk["offsets"], G6[prefix + "offsets"])
    assert np.array_equal(pk["input_ids"], G6[prefix + "input_ids"])
    assert np.array_equal(pk["token_type_ids"], G6[prefix + "token_type_ids"])
    for k in ("label", "rel_label", "pre_type", "q_head_idx", "a_head_idx", "sep_idx", "rel_idx"):
        if prefix + k in G6.files:
            assert np.array_equal(pk[k], G6[prefix + k]), k
    assert [f["head_ent"] or "" for f in ds.features] == G6[prefix + "head_ent"].tolist()
    assert [f["tail_ent"] or "" for f in ds.features] == G6[prefix + "tail_ent"].tolist()
    assert sorted(ds[0].keys()) == G6[prefix + "keys"].tolist()
    assert all(f["attention_mask"] == [1] * len(f["input_ids"]) for f in ds.features)


def test_finetune_features_config_and_collator(dirs):
    tok = BertWordPieceTokenizer(G6["vocab"].tolist())
    dm = KGC(_args(dirs), None, tokenizer=tok, visual_features=_vis())
    dm.setup()
    cfg = dm.get_config()
    for k in ("entity_id_st", "entity_id_ed", "relation_id_st", "relation_id_ed"):
        assert cfg[k] == int(G6["cfg_" + k])
    assert cfg["analogy_entity_ids"] == G6["cfg_analogy_entity_ids"].tolist()
    assert cfg["analogy_relation_ids"] == G6["cfg_analogy_relation_ids"].tolist()
    assert len(dm.tokenizer) == int(G6["len_tokenizer"]) == 30522 + 11292 + 192
    assert (cfg["entity_id_st"], cfg["relation_id_st"]) == (30522, 30522 + 11292)
    for split, ds in (("train", dm.data_train), ("dev", dm.data_val), ("test", dm.data_test)):
        _check_split(ds, f"ft_{split}_")
    assert len(dm.data_val) == 1020                                        # MARS dev in full
    # the reference collator output, field by field (pixel_values literal on the host, then the device-table index form)
    dm.use_host_pixels(True)
    for name, ds in (("dev8", dm.data_val), ("mix", dm.data_train)):
        rows = G6[f"col_{name}_rows"].tolist()
        before = copy.deepcopy([ds[i] for i in rows])
        batch = dm.sampler([ds[i] for i in rows])
        assert [ds[i] for i in rows] == before                              # feature dicts are not consumed
        keys = {k[len(f"col_{name}_"):] for k in G6.files if k.startswith(f"col_{name}_")} - {"rows"}
        assert keys == set(batch.keys())
        for k in keys:
            ref = G6[f"col_{name}_{k}"]
            got = batch[k].numpy() if torch.is_tensor(batch[k]) else np.array(batch[k])
            assert got.shape == ref.shape and np.array_equal(got, ref), k
    dm.use_host_pixels(False)
    rows = G6["col_mix_rows"].tolist()
    batch = dm.sampler([dm.data_train[i] for i in rows])
    assert "pixel_values" not in batch and batch["image_index"].dtype == torch.int32
    vis = _vis()
    rebuilt = torch.stack([torch.stack([vis[i] if i >= 0 else torch.zeros(3, 4, 4) for i in r]) for r in batch["image_index"].tolist()])
    assert np.array_equal(rebuilt.numpy(), G6["col_mix_pixel_values"])
    # loaders: shuffle only for train, eval batch size for the others
    b0 = next(iter(dm.val_dataloader()))
    assert b0["input_ids"].shape[0] == 8 and np.array_equal(b0["label"].numpy(), G6["ft_dev_label"][:8])


def test_pretrain_features_follow_the_module_rng(dirs):
    tok = BertWordPieceTokenizer(G6["vocab"].tolist())
    P.reseed(1)
    args = _args(dirs, pretrain=1, L=32)
    dm = KGC(args, None, tokenizer=tok, visual_features=_vis())
    fresh = dm._fresh_tokenizer()
    train = P.get_dataset(dm.args, dm.processor, "train", fresh)
    dev = P.get_dataset(dm.args, dm.processor, "dev", fresh)            # same triples, RNG stream continues
    _check_split(train, "pt_train_")
    _check_split(dev, "pt_dev_")
    assert not np.array_equal(train.packed()["input_ids"], dev.packed()["input_ids"])
    dm.use_host_pixels(True)
    batch = dm.sampler([train[i] for i in range(12)])
    for k in {k[len("col_pt_"):] for k in G6.files if k.startswith("col_pt_")}:
        got = batch[k].numpy() if torch.is_tensor(batch[k]) else np.array(batch[k])
        assert np.array_equal(got, G6["col_pt_" + k]), k
    # quirk kept (processor.py:290 with rel_id = -1): the [PAD] placeholder of pre_type 1 becomes 30522 + E - 1
    f1 = train[0]
    assert f1["pre_type"] == 1 and (30522 + 11292 - 1) in f1["input_ids"]


def test_cache_round_trip_and_reference_class_name(dirs, tmp_path):
    import pickle
    tok = BertWordPieceTokenizer(G6["vocab"].tolist())
    args = _args(dirs)
    args.overwrite_cache = False
    proc = P.KGProcessor(tok, args)
    path = P.cache_path(args, "test")
    assert path.endswith("cached_test_featuresbert-base-uncased_pretrain0.pkl")
    if os.path.exists(path):
        os.remove(path)
    a = P.get_dataset(args, proc, "test", tok)
    assert os.path.exists(path)
    b = P.get_dataset(args, proc, "test", tok)                          # served from the cache
    assert a.features == b.features
    # a cache written by the reference pickles data.processor.KGCDataset: emulate the class path
    import sys, types
    mod = types.ModuleType("data.processor")
    pkg = types.ModuleType("data")
    class KGCDataset:                                                    # noqa: E306
        def __init__(self, features): self.features = features
    KGCDataset.__module__, KGCDataset.__qualname__ = "data.processor", "KGCDataset"
    mod.KGCDataset = KGCDataset
    sys.modules["data"], sys.modules["data.processor"] = pkg, mod
    try:
        blob = pickle.dumps(KGCDataset(a.features))
    finally:
        del sys.modules["data"], sys.modules["data.processor"]
    with open(path, "wb") as f:
        f.write(blob)
    c = P.get_dataset(args, proc, "test", tok)
    assert isinstance(c, P.KGCDataset) and c.features == a.features
    os.remove(path)
