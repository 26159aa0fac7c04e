"""Host-side logic that needs no GPU: parameter names / flat layout / optimizer grouping / schedule / metrics / API surface."""
import argparse
import os

import numpy as np
import pytest
import torch

from oracle import mkgformer_oracle as O


def _tiny_model():
    from mkg_analogy_amd.models import MKGformerKGC, TextConfig, VisionConfig
    tc = TextConfig(vocab_size=100, max_position_embeddings=32)
    return MKGformerKGC(VisionConfig(patch_size=32), tc), tc


@pytest.fixture(scope="module")
def model():
    return _tiny_model()[0]


def test_parameter_names_match_reference(model):
    """451 tensors with the reference's names and shapes (the oracle table is pinned to the reference by G1/G4)."""
    ref = O.param_shapes(O.VisionCfg(patch_size=32), O.TextCfg(vocab_size=100, max_position_embeddings=32))
    got = {n: tuple(p.shape) for n, p in model.named_parameters()}
    assert len(got) == 451
    assert got == ref
    assert list(got) == list(ref), "named_parameters() order differs from the reference construction order"
    sd = model.state_dict()
    assert "cls.predictions.decoder.weight" in sd and "cls.predictions.decoder.bias" in sd
    assert "unimo.text_embeddings.position_ids" in sd and "unimo.vision_embeddings.position_ids" in sd
    assert len(sd) == 455


def test_tied_embeddings_and_resize(model):
    m, tc = _tiny_model()
    assert m.get_input_embeddings().weight is m.get_output_embeddings().weight
    old = m.get_input_embeddings().weight.detach().clone()
    b_old = m.cls.predictions.bias.detach().clone()
    m.resize_token_embeddings(107)
    w = m.get_input_embeddings().weight
    assert w.shape == (107, 768) and m.get_output_embeddings().weight is w
    assert torch.equal(w[:100], old)
    assert m.cls.predictions.bias.shape == (107,) and torch.equal(m.cls.predictions.bias[:100], b_old)
    assert float(m.cls.predictions.bias[100:].abs().max()) == 0.0
    assert m.cls.predictions.decoder.bias is m.cls.predictions.bias
    assert abs(float(w[100:].std()) - 0.02) < 0.01          # N(0, initializer_range) rows


def test_flat_layout_covers_every_parameter_once(model):
    from mkg_analogy_amd.params import ALIGN, DEAD, NO_DECAY, gemm_weight_names, layout_order
    names = [n for n, _ in model.named_parameters()]
    order = layout_order(12)
    assert sorted(order) == sorted(names) and len(set(order)) == len(order)
    # backward-completion order: head first, layer 11 before layer 0; the text stream's embedding tables (tied word embedding last of them) right
    # behind text layer 0 and in FRONT of vision layer 0 (they are final when the text stream ends: engine.backward releases them then); the vision
    # embeddings and the never-trained tensors at the very end
    assert order[0].startswith("cls.predictions.transform")
    assert order.index("unimo.encoder.text_layer.11.output.dense.weight") < order.index("unimo.encoder.vision_layers.11.mlp.fc1.weight") \
        < order.index("unimo.encoder.text_layer.10.output.dense.weight") < order.index("unimo.encoder.vision_layers.0.mlp.fc1.weight")
    iw = order.index("unimo.text_embeddings.word_embeddings.weight")
    assert order[iw - 1] == "cls.predictions.bias" and order[iw + 1] == "unimo.encoder.vision_layers.0.self_attn.q_proj.weight"
    assert order.index("unimo.encoder.text_layer.0.output.LayerNorm.bias") < iw < order.index("unimo.vision_embeddings.patch_embedding.weight")
    assert order[-1] == "unimo.text_pooler.dense.bias"
    # every GEMM weight group is made of adjacent, alignment-preserving members
    shapes = {n: tuple(p.shape) for n, p in model.named_parameters()}
    for key, members in gemm_weight_names(12):
        idx = [order.index(n) for n in members]
        assert idx == list(range(idx[0], idx[0] + len(idx))), key
        for n in members:
            assert (shapes[n][0] * shapes[n][1]) % ALIGN == 0
    # optimizer grouping quirk is the reference's substring rule
    for n in names:
        assert (0.0 if any(nd in n for nd in NO_DECAY) else 0.01) == O.decay_of(n)
    assert all(n.startswith(DEAD) for n in ("unimo.text_pooler.dense.weight", "unimo.vision_post_layernorm.bias"))


def test_linear_warmup_schedule_matches_reference_golden(golden_dir):
    from mkg_analogy_amd.optim import LinearWarmupSchedule
    g = np.load(os.path.join(golden_dir, "g4_adamw.npz"))
    T = int(g["num_training_steps"])

    class _Opt:
        param_groups = [{"lr": 5e-5}, {"lr": 5e-5}]
    opt = _Opt()
    s = LinearWarmupSchedule(opt, num_warmup_steps=0.1 * T, num_training_steps=T)
    lrs = []
    for _ in range(T + 1):
        lrs.append(opt.param_groups[0]["lr"])
        s.step()
    np.testing.assert_allclose(np.array(lrs) / 5e-5, g["sched_curve"], atol=1e-12)
    np.testing.assert_allclose(lrs[:3], g["lrs"], atol=1e-15)
    assert opt.param_groups[1]["lr"] == opt.param_groups[0]["lr"]


def test_epoch_end_metrics_match_reference_golden(golden_dir):
    from mkg_analogy_amd.lit_models.transformer import TransformerLitModel
    g = np.load(os.path.join(golden_dir, "g1_tiny_e2e.npz"))
    lit = TransformerLitModel.__new__(TransformerLitModel)
    torch.nn.Module.__init__(lit)
    lit.logged = {}
    lit.validation_epoch_end([{"entity_ranks": g["ranks"]}])
    for name, val in zip(g["metric_names"].tolist(), g["metric_vals"].tolist()):
        assert abs(lit.logged[name] - val) < 1e-6, name
    lit.logged = {}
    lit.test_epoch_end([{"entity_ranks": np.array([1, 4, 11])}, {"entity_ranks": np.array([21, 2])}, {"relation_ranks": np.array([1])}])
    assert lit.logged["Eval_entity/hits1"] == pytest.approx(0.2) and lit.logged["Eval_entity/hits10"] == pytest.approx(0.6)
    assert lit.logged["Eval_entity/mrr"] == pytest.approx(np.mean(1.0 / np.array([1, 4, 11, 21, 2])))


def test_trainer_surface_constructor_and_argparse():
    from mkg_analogy_amd import data_synth as D
    from mkg_analogy_amd.lit_models import TransformerLitModel
    from mkg_analogy_amd.models import MKGformerKGC
    p = argparse.ArgumentParser()
    MKGformerKGC.add_to_argparse(p)
    TransformerLitModel.add_to_argparse(p)
    a = p.parse_args([])
    assert a.pretrain == 0 and a.lr == 5e-5 and a.weight_decay == 0.01 and a.label_smoothing == 0.1 and a.optimizer == "AdamW"
    m, _ = _tiny_model()
    tok = D.FakeTokenizer(n=100)
    a.alpha, a.warm_up_radio = 0.43, 0.1
    lit = TransformerLitModel(model=m, args=a, tokenizer=tok, data_config=dict(entity_id_st=10, entity_id_ed=60, relation_id_st=60,
                                                                               relation_id_ed=100, analogy_entity_ids=[11, 12, 13],
                                                                               analogy_relation_ids=[61, 62, 65]))
    assert lit.entity_id_ed == 60 and lit.alpha == 0.43
    lit._init_relation_word()                                       # CPU-side surgery only; no GPU needed until forward
    w = m.get_input_embeddings().weight
    assert w.shape[0] == 101
    np.testing.assert_allclose(w[100].detach().numpy(), w[[61, 62, 65]].mean(0).detach().numpy(), atol=1e-7)
    assert m.get_input_embeddings().weight is m.get_output_embeddings().weight


def test_synthetic_batch_contract():
    from mkg_analogy_amd import data_synth as D
    b = D.make_batch(5, 64, seed=3)
    assert b["pixel_values"].shape == (5, 2, 3, 224, 224) and b["pixel_values"].dtype == torch.float32
    for i in range(5):
        ids = b["input_ids"][i]
        assert ids[0] == D.CLS and (ids == D.MASK).sum() == 1 and (ids == D.SEP).sum() == 6 and (ids == D.R_TOKEN).sum() == 2
        n = int(b["attention_mask"][i].sum())
        assert (ids[n:] == 0).all() and 40 <= n <= 64
        sep = b["sep_idx"][i]
        assert (ids[sep] == D.SEP).all() and (ids[b["rel_idx"][i]] == D.R_TOKEN).all()
        assert D.BASE_VOCAB <= ids[b["q_head_idx"][i]] < D.BASE_VOCAB + D.N_ENT
        assert D.BASE_VOCAB <= ids[b["a_head_idx"][i]] < D.BASE_VOCAB + D.N_ENT
        assert (b["token_type_ids"][i][: sep[2] + 1] == 0).all() and (b["token_type_ids"][i][sep[2] + 1:n] == 1).all()
    cfg = D.data_config()
    assert len(cfg["analogy_entity_ids"]) == 2063 and len(set(cfg["analogy_entity_ids"])) == 2063
    assert D.VOCAB == 42007 and D.R_TOKEN == 42006


def test_device_image_table_slot_rules():
    """Same slot selection as the reference collator (MarT/data/data_module.py:126-142), restated with explicit branches."""
    from mkg_analogy_amd.batching import DeviceImageTable
    ents = ["Q1", "Q2", "Q3", "Q2"]
    t = DeviceImageTable(ents)
    head = ["Q1", "Q3", None, None, "Q2", ""]
    tail = ["Q2", None, "Q3", None, "", "Q1"]
    got = t.slots(head, tail).tolist()
    exp = []
    for h, tl in zip(head, tail):
        if h and tl:
            exp.append([ents.index(h), ents.index(tl)])
        else:
            e = h if h is not None else tl
            exp.append([ents.index(e) if e else -1, -1])
    assert got == exp == [[0, 1], [2, -1], [2, -1], [-1, -1], [1, -1], [-1, -1]]


def test_flava_parameter_names_and_layout():
    """FlavaKGC container tree == the reference's named_parameters (order included; the oracle table is pinned by golden G5)."""
    from oracle import flava_oracle as FO
    from mkg_analogy_amd.flava_engine import FLAVA_DEAD, flava_gemm_groups, flava_layout_order
    from mkg_analogy_amd.models import FlavaKGC, flava_config
    c = FO.FlavaCfg(vocab_size=120, text_layers=2, image_layers=2, mm_layers=1, max_position_embeddings=32)
    m = FlavaKGC(flava_config(vocab_size=120, text_layers=2, image_layers=2, mm_layers=1, max_position_embeddings=32))
    ref = FO.param_shapes(c)
    got = {n: tuple(p.shape) for n, p in m.named_parameters()}
    assert got == ref and list(got) == list(ref)
    order = flava_layout_order(2, 2, 1)
    assert sorted(order) == sorted(ref) and len(set(order)) == len(order)
    for key, members in flava_gemm_groups(2, 2, 1):
        idx = [order.index(n) for n in members]
        assert idx == list(range(idx[0], idx[0] + len(idx))), key
    dead = FLAVA_DEAD((2, 2, 1))
    assert sum(1 for n in ref if n.startswith(dead)) == 1 + 1 + 4 + 6 + 4 + 2 * (2 + 1)      # == reference None-grad set rule (golden g5: 26 at 3/3/2)
    assert m.get_input_embeddings().weight is m.get_output_embeddings().weight
    m.resize_token_embeddings(125)
    assert m.cls.bias.shape == (125,) and m.get_output_embeddings().weight.shape == (125, 768)


def test_trainer_world_size_shards_and_tail_window():
    """Trainer host logic (no GPU): world size defaults to the process group's, a DistributedSampler-sharded loader is counted at
    its unsharded length for num_training_steps (lit_models/base.py:86-91 divides by the device count itself), and the last
    partial accumulation window of an epoch still steps the optimizer (as PL does)."""
    import torch
    from torch.utils.data import DataLoader, TensorDataset
    from torch.utils.data.distributed import DistributedSampler
    from mkg_analogy_amd.trainer import Trainer
    t = Trainer(max_epochs=2, accumulate_grad_batches=4)
    assert t.world_size == 1                                            # no process group here
    ds = TensorDataset(torch.arange(40))
    plain = DataLoader(ds, batch_size=4)
    sharded = DataLoader(ds, batch_size=4, sampler=DistributedSampler(ds, num_replicas=2, rank=0, shuffle=False))
    assert Trainer._shards(plain) == 1 and Trainer._shards(sharded) == 2
    assert len(sharded) * Trainer._shards(sharded) == len(plain)

    class _Opt:
        def __init__(self):
            self.steps, self.zeroed, self.param_groups, self.grad_scale = 0, 0, [{"lr": 0.0}], 1.0

        def zero_grad(self):
            self.zeroed += 1

        def step(self):
            self.steps += 1

    class _Sync:
        reducer, grad_scale = None, 1.0

        def begin(self):
            pass

        def finish(self):
            pass

    class _Eng:
        grad_ready = grad_ready_async = None

    class _Model:
        engine = _Eng()

        def train(self):
            pass

    class _Lit:
        model = _Model()

        def training_step(self, batch, i):
            return torch.zeros((), requires_grad=True)

    t.optimizer, t.scheduler, t.sync, t.stream_optimizer = _Opt(), type("S", (), {"step": lambda self: None})(), _Sync(), False
    n = 10                                                              # 10 batches, windows of 4: two full windows + a tail of 2
    for i in range(n):
        t.train_step(_Lit(), {}, i, end_of_epoch=(i == n - 1))
    assert t.optimizer.steps == 3 and t.global_step == 3


def test_collator_rejects_prompts_without_exactly_one_mask():
    import pytest
    import torch
    from mkg_analogy_amd.data.data_module import DataCollatorForSeq2Seq

    class _Tok:
        mask_token_id = 103

        def pad(self, feats, **kw):
            return {k: torch.tensor([f[k] for f in feats]) for k in feats[0]}

    col = DataCollatorForSeq2Seq(_Tok(), num_labels=10)
    ok = [dict(input_ids=[101, 5, 103, 102], attention_mask=[1] * 4, token_type_ids=[0] * 4, label=3)]
    assert col(ok)["label"].tolist() == [3]
    with pytest.raises(ValueError):
        col([dict(input_ids=[101, 5, 6, 102], attention_mask=[1] * 4, token_type_ids=[0] * 4, label=3)])      # [MASK] truncated away
    with pytest.raises(ValueError):
        col([dict(input_ids=[101, 103, 103, 102], attention_mask=[1] * 4, token_type_ids=[0] * 4, label=3)])
    with pytest.raises(ValueError):
        col([dict(input_ids=[101, 5, 103, 102], attention_mask=[1] * 4, token_type_ids=[0] * 4, label=-100)])  # ignore_index unsupported


def test_rank_seed_is_a_hash_of_base_seed_and_rank():
    """distributed.rank_seed: rank 0 keeps the single-process dropout stream, every other rank gets a distinct hashed base
    seed small enough that the model's per-step seed (base * 1000003 + step * 7919 + per-layer offsets < 64) never collides across
    (rank, step, offset) for realistic step counts."""
    from mkg_analogy_amd.distributed import rank_seed
    base = 0x5EED
    seeds = [rank_seed(base, r) for r in range(8)]
    assert seeds[0] == base and len(set(seeds)) == 8
    assert all(0 < s < (1 << 26) + 1 for s in seeds[1:])
    assert rank_seed(base, 3) == rank_seed(base, 3) and rank_seed(base + 1, 3) != rank_seed(base, 3)
    used = set()
    for s in seeds:
        for step in range(0, 2000, 7):
            for off in (0, 1, 10, 11, 12, 57):
                v = (s * 1000003 + step * 7919) & 0x7FFFFFFFFFFF
                assert (v + off) not in used
                used.add(v + off)


@pytest.mark.parametrize("cond", [False, True])
def test_seeded_weights_equal_the_golden_generators_recipe(model, cond):
    """bench.py regenerates the goldens' weight sets with data_synth.seeded_weights (the product never imports the oracle): same tensors, bit for
    bit, as oracle.init_params (+ condition_weights), which is what oracle/gen_goldens_full.py loaded into the unmodified reference."""
    from mkg_analogy_amd import data_synth as D
    ref = O.init_params(O.VisionCfg(patch_size=32), O.TextCfg(vocab_size=100, max_position_embeddings=32), seed=3)
    if cond:
        ref = O.condition_weights(ref)
    got = D.seeded_weights([(n, p.shape) for n, p in model.named_parameters()], seed=3, conditioned=cond)
    assert list(got) == list(ref)
    for n in ref:
        assert torch.equal(got[n], ref[n]), n


def test_scored_ids_uniqueness_check_is_per_tensor_object_not_per_address():
    """functional._require_unique (round-3 advisor finding): the cache must never answer for a DIFFERENT id tensor that happens to live at the
    address of an earlier one (the caching allocator recycles addresses): keyed on the tensor object and its in-place version."""
    from mkg_analogy_amd import functional as Fn
    good = torch.tensor([3, 1, 2], dtype=torch.int32)
    Fn._require_unique(good)
    Fn._require_unique(good)                                  # cached: same object, same version
    good[0] = 1                                               # in-place edit bumps _version -> checked again
    with pytest.raises(ValueError):
        Fn._require_unique(good)
    for _ in range(200):                                      # fresh objects (recycled ids / addresses included) are always checked
        bad = torch.tensor([5, 5, 7], dtype=torch.int32)
        with pytest.raises(ValueError):
            Fn._require_unique(bad)
        ok = torch.tensor([5, 6, 7], dtype=torch.int32)
        Fn._require_unique(ok)
    rng = torch.arange(0, 10, dtype=torch.int32)
    rng._mart_unique = True                                   # LazyRows marks ranges: no check, no device sync
    Fn._require_unique(rng)
    assert len(Fn._UNIQUE_OK) <= 66
