"""G6: MARS / MarKG prompt + feature plumbing goldens, by IMPORTING THE UNMODIFIED REFERENCE data package.

    python oracle/gen_goldens_data.py        # writes tests/golden/g6_mars_plumbing.npz

TEST INFRASTRUCTURE ONLY (build container only; /root/reference does not exist on the GPU box).  The reference's
``data.processor`` / ``data.data_module`` are imported in place and run on the data fixtures under tests/golden/mars/
(MARS dev.json in full, the first 200 lines of train/test.json, the analogy entity / relation lists, MarKG
entity2text.txt + relation2text.txt, the first 400 MarKG triples); only inputs and outputs are stored.

Shims (installed libraries / sys.modules only, never reference files):
  3. stub ``pytorch_lightning`` (LightningDataModule), as in gen_goldens.py
  5. ``AutoTokenizer.from_pretrained`` -> the installed ``transformers.BertTokenizer`` over a deterministic synthetic
     30 522-entry vocabulary (mkg_analogy_amd.data_synth.synthetic_wordpiece_vocab; the real vocab.txt is a download).
     The reference asks for ``use_fast=False`` (transformers 4.19 slow tokenizer); the installed 5.x class runs the same
     published BERT algorithm on the ``tokenizers`` backend.  The two differ only in how ``longest_first`` splits an ODD
     excess between two over-long inputs; the fixtures never truncate both inputs (asserted below).
  6. a small random ``entity_image_features.CLIP-VIT-16-32.pth`` ([N,3,4,4]) in the temporary data_dir.
"""
from __future__ import annotations

import argparse
import copy
import os
import shutil
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = "/root/reference/MarT"
FIX = os.path.join(ROOT, "tests", "golden", "mars")


def fixture_texts():
    texts = []
    for fn in ("entity2text.txt", "relation2text.txt"):
        with open(os.path.join(FIX, fn), encoding="utf-8") as f:
            for line in f:
                texts.append(line.split("\t", 1)[1][:-1])
    return texts


def make_dirs(tmp):
    data_dir, pre = os.path.join(tmp, "MARS"), os.path.join(tmp, "MarKG")
    shutil.rmtree(tmp, ignore_errors=True)
    os.makedirs(data_dir); os.makedirs(pre)
    for fn in ("dev.json", "train.json", "test.json", "analogy_entities.txt", "analogy_relations.txt"):
        shutil.copy(os.path.join(FIX, fn), data_dir)
    for fn in ("entity2text.txt", "relation2text.txt", "wiki_tuple_ids.txt"):
        shutil.copy(os.path.join(FIX, fn), pre)
    return data_dir, pre


def pack(features, prefix, out):
    lens = np.array([len(f["input_ids"]) for f in features])
    out[prefix + "offsets"] = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    for k in ("input_ids", "attention_mask", "token_type_ids"):
        out[prefix + k] = np.array([t for f in features for t in f[k]], dtype=np.int32)
    for k in ("label", "rel_label", "pre_type", "q_head_idx", "a_head_idx", "sep_idx", "rel_idx"):
        if k in features[0]:
            out[prefix + k] = np.array([f[k] for f in features], dtype=np.int64)
    for k in ("head_ent", "tail_ent"):
        out[prefix + k] = np.array([f[k] or "" for f in features])
    out[prefix + "keys"] = np.array(sorted(features[0].keys()))


def main():
    import transformers
    from transformers import BertTokenizer
    from mkg_analogy_amd.data_synth import synthetic_wordpiece_vocab
    vocab = synthetic_wordpiece_vocab(fixture_texts())
    vmap = {w: i for i, w in enumerate(vocab)}

    def from_pretrained(name, *a, **k):                                               # shim 5
        return BertTokenizer(vocab=dict(vmap))
    import transformers.models.auto.tokenization_auto as ta
    ta.AutoTokenizer.from_pretrained = staticmethod(from_pretrained)
    transformers.AutoTokenizer.from_pretrained = staticmethod(from_pretrained)
    pl = types.ModuleType("pytorch_lightning")                                        # shim 3

    class LightningDataModule:
        def __init__(self, *a, **k):
            pass
    pl.LightningDataModule = LightningDataModule
    pl.LightningModule = torch.nn.Module
    sys.modules["pytorch_lightning"] = pl
    sys.path.insert(0, REF)
    import data.data_module as rdm                                                   # the reference, in place
    import data.processor as rproc

    tmp = os.path.join(ROOT, "gpurun_out", "_golden_tmp")
    data_dir, pre = make_dirs(tmp)
    n_ent = sum(1 for _ in open(os.path.join(pre, "entity2text.txt")))
    g = torch.Generator().manual_seed(7)
    vis = torch.randn(n_ent, 3, 4, 4, generator=g)
    torch.save(vis, os.path.join(data_dir, "entity_image_features.CLIP-VIT-16-32.pth"))  # shim 6

    out = {"vocab": np.array(vocab), "visual_seed": np.array(7)}
    L = 64
    args = argparse.Namespace(model_name_or_path="bert-base-uncased", data_dir=data_dir, pretrain_path=pre, pretrain=0,
                              max_seq_length=L, overwrite_cache=True, precision=32, model_class="MKGformerKGC",
                              batch_size=8, eval_batch_size=8, num_workers=0)
    # ---------------- fine-tune
    dm = rdm.KGC(args, None)
    dm.setup()
    cfg = dm.get_config()
    for k in ("entity_id_st", "entity_id_ed", "relation_id_st", "relation_id_ed"):
        out["cfg_" + k] = np.array(cfg[k])
    out["cfg_analogy_entity_ids"] = np.array(cfg["analogy_entity_ids"])
    out["cfg_analogy_relation_ids"] = np.array(cfg["analogy_relation_ids"])
    out["cfg_keys"] = np.array(sorted(k for k in cfg if not k.startswith("data_")))
    out["len_tokenizer"] = np.array(len(dm.tokenizer))
    for split, ds in (("train", dm.data_train), ("dev", dm.data_val), ("test", dm.data_test)):
        feats = [copy.deepcopy(ds[i]) for i in range(len(ds))]
        assert max(len(f["input_ids"]) for f in feats) <= L
        pack(feats, f"ft_{split}_", out)
    # collated batches: dev rows 0..7 and a mode-mixed pick of train rows
    for name, ds, rows in (("dev8", dm.data_val, list(range(8))), ("mix", dm.data_train, [0, 17, 41, 80, 123, 150, 199])):
        batch = dm.sampler([copy.deepcopy(ds[i]) for i in rows])
        out[f"col_{name}_rows"] = np.array(rows)
        for k, v in batch.items():
            out[f"col_{name}_{k}"] = v.numpy() if torch.is_tensor(v) else np.array(v)
    # ---------------- pre-train (train then dev: the module RNG keeps running across splits)
    args.pretrain = 1
    args.max_seq_length = 32            # forces longest_first truncation of the single input on long relation texts
    dmp = rdm.KGC(args, None)
    dmp.data_train = rproc.get_dataset(args, dmp.processor, "train")
    dmp.data_val = rproc.get_dataset(args, dmp.processor, "dev")
    for split, ds in (("train", dmp.data_train), ("dev", dmp.data_val)):
        feats = [copy.deepcopy(ds[i]) for i in range(len(ds))]
        pack(feats, f"pt_{split}_", out)
    batch = dmp.sampler([copy.deepcopy(dmp.data_train[i]) for i in range(12)])
    for k, v in batch.items():
        out[f"col_pt_{k}"] = v.numpy() if torch.is_tensor(v) else np.array(v)
    path = os.path.join(ROOT, "tests", "golden", "g6_mars_plumbing.npz")
    np.savez_compressed(path, **out)
    shutil.rmtree(tmp, ignore_errors=True)
    print("wrote", path, os.path.getsize(path), "bytes;", {k: v.shape for k, v in out.items() if k.startswith("ft_dev_")})


if __name__ == "__main__":
    main()
