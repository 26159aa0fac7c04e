"""BASELINE configs[0] as a parity case: MARS-dev through the real prompt pipeline (tests/golden/mars fixtures, synthetic
WordPiece vocabulary), device-resident image table, the HIP MKGformer path and the eval ranking -- against the CPU
oracle on the same batches.  Real text/vision widths (768, 12+12 layers); 32x32 images keep the oracle in seconds."""
import argparse
import os
import shutil

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mkgformer_oracle as O  # noqa: E402  (checker only)

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = os.path.join(HERE, "golden", "mars")
IMG, PATCH = 32, 16


@pytest.fixture(scope="module")
def world(tmp_path_factory):
    from mkg_analogy_amd.data import BertWordPieceTokenizer
    from mkg_analogy_amd.data.data_module import KGC
    from mkg_analogy_amd.data_synth import synthetic_wordpiece_vocab
    from mkg_analogy_amd.lit_models import TransformerLitModel
    from mkg_analogy_amd.models import MKGformerKGC, TextConfig, VisionConfig
    tmp = tmp_path_factory.mktemp("mars")
    data_dir, pre = tmp / "MARS", tmp / "MarKG"
    data_dir.mkdir(); pre.mkdir()
    for fn in ("dev.json", "train.json", "test.json", "analogy_entities.txt", "analogy_relations.txt"):
        shutil.copy(os.path.join(FIX, fn), data_dir)
    for fn in ("entity2text.txt", "relation2text.txt", "wiki_tuple_ids.txt"):
        shutil.copy(os.path.join(FIX, fn), pre)
    texts = []
    for fn in ("entity2text.txt", "relation2text.txt"):
        with open(os.path.join(FIX, fn), encoding="utf-8") as f:
            texts += [line.split("\t", 1)[1][:-1] for line in f]
    tok = BertWordPieceTokenizer(synthetic_wordpiece_vocab(texts))
    args = argparse.Namespace(model_name_or_path="bert-base-uncased", data_dir=str(data_dir), pretrain_path=str(pre), pretrain=0,
                              max_seq_length=64, overwrite_cache=True, precision=16, model_class="MKGformerKGC", batch_size=8,
                              eval_batch_size=8, num_workers=0, label_smoothing=0.1, alpha=0.43, lr=5e-5, weight_decay=0.01,
                              optimizer="AdamW", warm_up_radio=0.1)
    n_ent = len(texts) - 192
    vis = torch.randn(n_ent, 3, IMG, IMG, generator=torch.Generator().manual_seed(5))
    dm = KGC(args, None, tokenizer=tok, visual_features=vis)
    dm.setup()
    torch.manual_seed(0)
    model = MKGformerKGC(VisionConfig(image_size=IMG, patch_size=PATCH), TextConfig())
    lit = TransformerLitModel(model=model, args=args, tokenizer=dm.tokenizer, data_config=dm.get_config())
    vc = O.VisionCfg(image_size=IMG, patch_size=PATCH)
    sd = O.init_params(vc, O.TextCfg(vocab_size=len(dm.tokenizer)), seed=21)
    for l in range(8, 12):                     # conditioned weights, see tests/test_model_gpu.py::_condition
        for k in ("weight", "bias"):
            n = f"unimo.encoder.text_layer.{l}.attention.self.value.{k}"
            sd[n] = sd[n] * 0.05
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected
    model.cuda()
    lit._init_relation_word()
    sd = O.init_relation_word(sd, dm.analogy_relation_ids)
    dm.attach(model)
    return dm, model, lit, vc, sd, vis


def _oracle_logits(world, batch, pixel_values):
    dm, model, lit, vc, sd, vis = world
    tc = O.TextCfg(vocab_size=len(dm.tokenizer))
    with torch.no_grad():
        _, trans = O.forward(sd, vc, tc, batch["input_ids"], batch["attention_mask"], batch["token_type_ids"], pixel_values,
                             batch["sep_idx"], train=False)
        B = batch["input_ids"].shape[0]
        _, mi = (batch["input_ids"] == 103).nonzero(as_tuple=True)
        return O.score(sd, trans[torch.arange(B), mi], dm.analogy_entity_ids), trans


def _pixels(vis, image_index):
    zero = torch.zeros_like(vis[0])
    return torch.stack([torch.stack([vis[i] if i >= 0 else zero for i in row]) for row in image_index.tolist()])


@pytest.mark.parametrize("first", [0, 392, 792])
def test_mars_dev_eval_batches(world, first):
    """Dev rows first..first+7 (392 / 792 straddle the mode 0->1 and 1->2 boundaries of dev.json)."""
    dm, model, lit, vc, sd, vis = world
    feats = [dm.data_val[i] for i in range(first, first + 8)]
    batch = dm.sampler(feats)
    assert "image_index" in batch and batch["input_ids"].shape[1] % 8 == 0        # precision 16 -> padded to 8 (data_module.py:213)
    ref_logits, _ = _oracle_logits(world, batch, _pixels(vis, batch["image_index"]))
    ref_ranks = O.ranks_double_sort(ref_logits, batch["label"])
    model.eval()
    out = lit.validation_step(dict(batch), 0)
    got_ranks = out["entity_ranks"]
    # logits of the HIP path on the same rows
    o, _ = model(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], token_type_ids=batch["token_type_ids"],
                 image_index=batch["image_index"], sep_idx=batch["sep_idx"], return_dict=True)
    rows = o.logits.mask_rows(batch["input_ids"].cuda(), 103)
    got_logits = rows[:, torch.tensor(dm.analogy_entity_ids, dtype=torch.int32, device="cuda")].float().cpu()
    scale = max(1.0, float(ref_logits.abs().max()))
    err = float((got_logits - ref_logits).abs().max())
    print(f"\nrows {first}..{first + 7}: L={batch['input_ids'].shape[1]} max|dlogit| {err:.3e} (tol {1e-2 * scale:.3e})  ranks hip {got_ranks.tolist()} oracle {ref_ranks.tolist()}")
    assert err < 1e-2 * scale
    # ranked index: exact wherever the label's logit is separated from its neighbours by more than the error
    lab = ref_logits[torch.arange(8), batch["label"]]
    lo = 1 + (ref_logits > (lab + 2 * err)[:, None]).sum(1).numpy()
    hi = 1 + (ref_logits > (lab - 2 * err)[:, None]).sum(1).numpy()
    assert np.all(got_ranks >= lo) and np.all(got_ranks <= hi)
    exact = lo == hi
    assert np.array_equal(got_ranks[exact], np.asarray(ref_ranks)[exact])


def test_mars_dev_text_only_and_metrics(world):
    """No images at all (configs[0] 'text-only ... no images'): image_index = -1 everywhere == zero pixel_values."""
    from mkg_analogy_amd.trainer import Trainer
    dm, model, lit, vc, sd, vis = world
    batches = []
    for first in (0, 8, 400):
        b = dm.sampler([dm.data_val[i] for i in range(first, first + 8)])
        b["image_index"] = torch.full_like(b["image_index"], -1)
        batches.append(b)
    ref_ranks = []
    for b in batches:
        lg, _ = _oracle_logits(world, b, torch.zeros(8, 2, 3, IMG, IMG))
        ref_ranks.append(np.asarray(O.ranks_double_sort(lg, b["label"])))
    ref = O.rank_metrics(np.concatenate(ref_ranks))
    got = Trainer().validate(lit, batches)
    print("\nmetrics hip", {k: round(v, 4) for k, v in got.items()}, "\nmetrics oracle", {k: round(float(v), 4) for k, v in ref.items()})
    assert set(ref) <= set(got)
    assert abs(got["Eval_entity/mean_rank"] - ref["Eval_entity/mean_rank"]) <= 0.02 * ref["Eval_entity/mean_rank"] + 1.0
    assert abs(got["Eval_entity/hits10"] - ref["Eval_entity/hits10"]) <= 2 / 24 + 1e-9


def test_mars_train_step_loss(world):
    """One fine-tune step on MARS train rows with all three modes; loss vs oracle (eval mode: dropout off)."""
    dm, model, lit, vc, sd, vis = world
    rows = [0, 17, 41, 80, 123, 150, 199, 60]
    batch = dm.sampler([dm.data_train[i] for i in rows])
    tc = O.TextCfg(vocab_size=len(dm.tokenizer))
    with torch.no_grad():
        _, trans = O.forward(sd, vc, tc, batch["input_ids"], batch["attention_mask"], batch["token_type_ids"],
                             _pixels(vis, batch["image_index"]), batch["sep_idx"], train=False)
        loss_ref, _ = O.finetune_loss(sd, trans, batch["input_ids"], batch["label"], batch["rel_idx"], batch["q_head_idx"],
                                      batch["a_head_idx"], torch.tensor(dm.analogy_entity_ids), alpha=0.43)
    model.eval()
    model.store.zero_grad()
    loss = lit.training_step(dict(batch), 1)
    loss.backward()
    torch.cuda.synchronize()
    print(f"\nMARS train batch: loss hip {float(loss):.5f} oracle {float(loss_ref):.5f}")
    assert abs(float(loss) - float(loss_ref)) < 6e-3
    g = model.store.g("unimo.text_embeddings.word_embeddings.weight")
    used = torch.unique(batch["input_ids"]).cuda()
    assert torch.isfinite(g).all() and float(g[used].abs().sum()) > 0
