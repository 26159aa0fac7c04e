"""Generate golden vectors by IMPORTING THE UNMODIFIED REFERENCE from /root/reference.

Run in the build container only (the reference does not exist on the GPU box):

    python oracle/gen_goldens.py            # writes tests/golden/*.npz

TEST INFRASTRUCTURE ONLY.  Nothing here is shipped or timed.  No reference source is
copied: the reference modules are loaded in place (sys.path / importlib), driven with
weights regenerated from a numpy seed (oracle.mkgformer_oracle.init_params), and only
inputs + outputs are stored.

Shims (applied to installed libraries / sys.modules, never to reference files):
  1. transformers.modeling_utils.apply_chunking_to_forward (moved in transformers 5.x)
  2. text_config.torchscript = False (attribute removed in 5.x; read at modeling_unimo.py:910)
  3. stub ``pytorch_lightning`` module (LightningModule = nn.Module + no-op hooks)
  4. fake tokenizer (no BERT vocab on disk)
"""
from __future__ import annotations

import argparse
import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import mkgformer_oracle as O  # noqa: E402

REF = "/root/reference/MarT"


def load_reference():
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu
    mu.apply_chunking_to_forward = pu.apply_chunking_to_forward                       # shim 1
    pl = types.ModuleType("pytorch_lightning")                                        # shim 3

    class LightningModule(torch.nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, name, val, **k):
            self.__dict__.setdefault("_logged", {})[name] = float(val)

    class LightningDataModule:
        def __init__(self, *a, **k):
            pass

    pl.LightningModule, pl.LightningDataModule = LightningModule, LightningDataModule
    sys.modules["pytorch_lightning"] = pl
    sys.path.insert(0, REF)
    lit = importlib.import_module("lit_models")
    spec = importlib.util.spec_from_file_location("ref_unimo", os.path.join(REF, "models/modeling_unimo.py"))
    unimo = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(unimo)
    return lit, unimo


class FakeTokenizer:                                                                   # shim 4
    mask_token_id = 103

    def __init__(self, n):
        self.n = n
        self.extra = {}

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
        return ["" for _ in ids]


def hf_configs(vc: O.VisionCfg, tc: O.TextCfg):
    from transformers import BertConfig, CLIPVisionConfig
    t = BertConfig(vocab_size=tc.vocab_size, hidden_size=tc.hidden_size, num_hidden_layers=tc.num_hidden_layers,
                   num_attention_heads=tc.num_attention_heads, intermediate_size=tc.intermediate_size,
                   max_position_embeddings=tc.max_position_embeddings)
    t.torchscript = False                                                              # shim 2
    v = CLIPVisionConfig(hidden_size=vc.hidden_size, intermediate_size=vc.intermediate_size,
                         num_hidden_layers=vc.num_hidden_layers, num_attention_heads=vc.num_attention_heads,
                         image_size=vc.image_size, patch_size=vc.patch_size)
    v.device = "cpu"
    return v, t


def load_into(model, sd):
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    allowed = ("position_ids", "cls.predictions.decoder.weight", "cls.predictions.decoder.bias")
    assert all(any(a in m for a in allowed) for m in missing), missing


# ----------------------------------------------------------------------------- tiny config used by G1/G4
TINY_V = O.VisionCfg(hidden_size=64, num_hidden_layers=12, num_attention_heads=4, intermediate_size=128,
                     image_size=64, patch_size=32)
TINY_BASE, TINY_E, TINY_R = 300, 40, 20


def tiny_text_cfg(vocab):
    return O.TextCfg(vocab_size=vocab, hidden_size=64, num_hidden_layers=12, num_attention_heads=4,
                     intermediate_size=128, max_position_embeddings=64)


def tiny_batch(B=3, L=24, seed=5):
    b = O.synthetic_batch(B, L, TINY_V, seed=seed, n_entities=TINY_E, n_analogy=17, base_vocab=TINY_BASE, n_rel=TINY_R)
    return b


def g1_g4(lit, unimo, out_dir):
    """G1: tiny end-to-end (12 layers so the idx>=7/>=8 wiring is exercised), eval mode.
    G4: AdamW group membership + 3 optimizer steps on the same model."""
    import argparse as ap
    vocab0 = TINY_BASE + TINY_E + TINY_R                       # 360; [R] is added by _init_relation_word
    tc = tiny_text_cfg(TINY_BASE)
    hv, ht = hf_configs(TINY_V, tc)
    torch.manual_seed(0)
    model = unimo.UnimoForMaskedLM(hv, ht)
    tok = FakeTokenizer(vocab0)
    batch = tiny_batch()
    analogy_rel = list(range(TINY_BASE + TINY_E, TINY_BASE + TINY_E + 7))
    data_cfg = dict(entity_id_st=TINY_BASE, entity_id_ed=TINY_BASE + TINY_E,
                    relation_id_st=TINY_BASE + TINY_E, relation_id_ed=TINY_BASE + TINY_E + TINY_R,
                    analogy_entity_ids=batch["analogy_entity_ids"].tolist(), analogy_relation_ids=analogy_rel)
    args = ap.Namespace(label_smoothing=0.1, alpha=0.43, pretrain=0, lr=5e-5, weight_decay=0.01,
                        optimizer="AdamW", warm_up_radio=0.1)
    lm = lit.TransformerLitModel(model=model, args=args, tokenizer=tok, data_config=data_cfg)   # resize -> 360
    sd0 = O.init_params(TINY_V, tiny_text_cfg(vocab0), seed=11)
    load_into(model, sd0)
    lm._init_relation_word()                                    # -> 361, [R] row = mean of analogy relation rows
    assert model.get_input_embeddings().weight.shape[0] == vocab0 + 1
    r_row = model.get_input_embeddings().weight[vocab0].detach().numpy().copy()
    model.eval()

    def clone_batch():
        keep = ("input_ids", "attention_mask", "token_type_ids", "pixel_values", "sep_idx", "rel_idx",
                "q_head_idx", "a_head_idx", "label", "rel_label")
        return {k: batch[k].clone() for k in keep}

    # the [R] token of the synthetic batch must be the id the reference assigned
    r_tok = tok.extra["[R]"]
    assert r_tok == vocab0 and int(batch["input_ids"][0, batch["rel_idx"][0, 0]]) == r_tok

    # forward (full logits, as the reference returns them)
    b = clone_batch()
    for k in ("label", "rel_label", "rel_idx", "q_head_idx", "a_head_idx"):
        b.pop(k)
    with torch.no_grad():
        out, trans = model(**b, return_dict=True)
    logits = out.logits
    B = logits.shape[0]
    _, mask_idx = (batch["input_ids"] == 103).nonzero(as_tuple=True)
    mask_rows = logits[torch.arange(B), mask_idx]

    # training_step loss + grads (eval mode => dropout off, deterministic)
    model.zero_grad()
    loss = lm.training_step(clone_batch(), 1)
    loss.backward()
    named = dict(model.named_parameters())
    grad_names = ["unimo.encoder.text_layer.3.attention.self.adaptive_weight.0",
                  "unimo.encoder.text_layer.9.attention.self.adaptive_weight.0",
                  "unimo.encoder.text_layer.9.attention.self.adaptive_weight.1",
                  "unimo.text_embeddings.word_embeddings.weight",
                  "unimo.vision_embeddings.patch_embedding.weight",
                  "unimo.vision_embeddings.class_embedding",
                  "unimo.vision_embeddings.position_embedding.weight",
                  "unimo.encoder.vision_layers.0.self_attn.q_proj.weight",
                  "unimo.encoder.vision_layers.10.self_attn.k_proj.bias",
                  "unimo.encoder.vision_layers.11.mlp.fc2.weight",
                  "unimo.encoder.text_layer.7.attention.self.key.weight",
                  "unimo.encoder.text_layer.7.attention.self.value.bias",
                  "unimo.encoder.text_layer.10.intermediate.fusion_dense.weight",
                  "unimo.encoder.text_layer.11.output.LayerNorm.weight",
                  "unimo.encoder.text_layer.11.attention.self.key.weight",
                  "cls.predictions.bias", "cls.predictions.transform.dense.weight"]
    grads = {"grad::" + n: named[n].grad.detach().numpy().copy() for n in grad_names}
    none_grad = sorted(n for n, p in named.items() if p.grad is None)
    grad_norms = {n: float(p.grad.norm()) for n, p in named.items() if p.grad is not None}

    # eval ranks + metrics
    ev = lm._eval(clone_batch(), 0)
    lm.validation_epoch_end([ev])
    metrics = dict(lm.__dict__["_logged"])

    # G4: optimizer groups + 3 steps
    class _Trainer:
        pass
    cfg_names = {}
    no_decay = ["bias", "LayerNorm.weight"]
    for n, _p in model.named_parameters():
        cfg_names[n] = 0.0 if any(nd in n for nd in no_decay) else 0.01
    # reproduce configure_optimizers without a PL trainer: it only needs num_training_steps
    type(lm).num_training_steps = property(lambda self: 50)
    oc = lm.configure_optimizers()
    opt, sched = oc["optimizer"], oc["lr_scheduler"]["scheduler"]
    ref_groups = {}
    name_of = {id(p): n for n, p in model.named_parameters()}
    for g in opt.param_groups:
        for p in g["params"]:
            ref_groups[name_of[id(p)]] = float(g["weight_decay"])
    assert ref_groups == cfg_names
    lrs = []
    losses = []
    for step in range(3):
        opt.zero_grad()
        l = lm.training_step(clone_batch(), 1)
        l.backward()
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sched.step()
        losses.append(float(l))
    after = {"after::" + n: named[n].detach().numpy().copy() for n in
             ["unimo.encoder.text_layer.9.attention.self.adaptive_weight.0",
              "unimo.encoder.vision_layers.0.layer_norm1.weight",
              "unimo.encoder.vision_layers.0.self_attn.q_proj.bias",
              "unimo.encoder.text_layer.11.output.dense.weight",
              "cls.predictions.transform.LayerNorm.bias",
              "unimo.vision_embeddings.class_embedding"]}
    sched_curve = [float(sched.lr_lambdas[0](s)) for s in range(0, 51)]

    np.savez_compressed(
        os.path.join(out_dir, "g1_tiny_e2e.npz"),
        **{"in::" + k: v.numpy() for k, v in batch.items()},
        weight_seed=np.int64(11), vocab0=np.int64(vocab0),
        analogy_relation_ids=np.array(analogy_rel),
        r_row=r_row,
        mask_rows=mask_rows.numpy(), trans=trans.numpy(), logits_b0=logits[0].numpy(),
        loss=np.float64(float(loss)), ranks=np.asarray(ev["entity_ranks"]),
        metric_names=np.array(sorted(metrics)), metric_vals=np.array([metrics[k] for k in sorted(metrics)]),
        none_grad=np.array(none_grad),
        grad_norm_names=np.array(sorted(grad_norms)), grad_norm_vals=np.array([grad_norms[k] for k in sorted(grad_norms)]),
        **grads,
    )
    np.savez_compressed(
        os.path.join(out_dir, "g4_adamw.npz"),
        group_names=np.array(sorted(ref_groups)), group_wd=np.array([ref_groups[k] for k in sorted(ref_groups)]),
        lrs=np.array(lrs), losses=np.array(losses), sched_curve=np.array(sched_curve),
        num_training_steps=np.int64(50), **after,
    )
    print("G1 loss", float(loss), "ranks", ev["entity_ranks"], "G4 losses", losses)


def g2(unimo, out_dir):
    """Real-dims per-op: one BertLayer (fusion + K/V export mode, with reweight + padding mask) and one
    CLIPEncoderLayer (with text K/V prefix) at H=768.  Weights regenerated from the numpy seed."""
    vc, tc = O.VisionCfg(patch_size=32), O.TextCfg(vocab_size=1000)
    hv, ht = hf_configs(vc, tc)
    sd = O.init_params(vc, tc, seed=21)
    rng = np.random.default_rng(77)
    B, L, Nv = 2, 64, 1 + 2 * vc.num_patches
    x_t = torch.from_numpy(rng.standard_normal((B, L, 768), dtype=np.float32))
    x_v = torch.from_numpy(rng.standard_normal((B, Nv, 768), dtype=np.float32))
    am = torch.ones(B, L, dtype=torch.long)
    am[0, 50:] = 0
    am[1, 61:] = 0
    sep = torch.tensor([[5, 7, 20, 33, 35, 37], [9, 11, 31, 44, 46, 48]])
    bl = unimo.BertLayer(ht).eval()
    cl = unimo.CLIPEncoderLayer(hv).eval()
    bl.load_state_dict({k[len("unimo.encoder.text_layer.9."):]: v for k, v in sd.items() if k.startswith("unimo.encoder.text_layer.9.")})
    cl.load_state_dict({k[len("unimo.encoder.vision_layers.9."):]: v for k, v in sd.items() if k.startswith("unimo.encoder.vision_layers.9.")})
    ext = unimo.get_extended_attention_mask(am, am.shape, "cpu")
    x_t.requires_grad_(True)
    x_v.requires_grad_(True)
    outs = bl(x_t, attention_mask=ext, visual_hidden_state=x_v, output_qks=True, sep_idx=sep)
    y_t, (k, v) = outs[0], outs[-1]
    y_v = cl(x_v, past_key_values=(k, v))[0]
    # a scalar that touches everything, for gradient goldens
    w_t = torch.from_numpy(rng.standard_normal(y_t.shape, dtype=np.float32))
    w_v = torch.from_numpy(rng.standard_normal(y_v.shape, dtype=np.float32))
    s = (y_t * w_t).sum() + (y_v * w_v).sum()
    s.backward()
    np.savez_compressed(
        os.path.join(out_dir, "g2_layers_768.npz"),
        # inputs x_t, x_v, w_t, w_v are regenerated from input_seed (same draw order) by the tests;
        # outputs are stored on a row subset to keep the fixture small
        weight_seed=np.int64(21), input_seed=np.int64(77), attention_mask=am.numpy(), sep_idx=sep.numpy(),
        x_t_sum=np.float64(float(x_t.detach().double().sum())), x_v_sum=np.float64(float(x_v.detach().double().sum())),
        y_t=y_t.detach().numpy()[:, ::4], y_v=y_v.detach().numpy()[:, ::8],
        k=k.detach().numpy()[:, ::7], v=v.detach().numpy()[:, ::7],
        gx_t=x_t.grad.numpy()[:, ::4], gx_v=x_v.grad.numpy()[:, ::8],
        g_w0=bl.attention.self.adaptive_weight[0].grad.numpy(), g_w1=bl.attention.self.adaptive_weight[1].grad.numpy(),
        g_key_w=bl.attention.self.key.weight.grad.numpy()[:8], g_fd_b=bl.intermediate.fusion_dense.bias.grad.numpy(),
        g_vq_b=cl.self_attn.q_proj.bias.grad.numpy(), g_vfc1_w=cl.mlp.fc1.weight.grad.numpy()[:8],
    )
    print("G2 ok", float(s))


def g3(lit, out_dir):
    """LabelSmoothSoftmaxCEV1, relaxation loss and double-sort ranks on fixed logits (incl. a tie row)."""
    import torch.nn.functional as F
    rng = np.random.default_rng(3)
    logits = torch.from_numpy(rng.standard_normal((6, 37), dtype=np.float32) * 3)
    label = torch.from_numpy(rng.integers(0, 37, size=6))
    lg = logits.clone().requires_grad_(True)
    loss = lit.utils.LabelSmoothSoftmaxCEV1(lb_smooth=0.1)(lg, label) if hasattr(lit, "utils") else None
    if loss is None:
        from lit_models.utils import LabelSmoothSoftmaxCEV1
        loss = LabelSmoothSoftmaxCEV1(lb_smooth=0.1)(lg, label)
    loss.backward()
    _, o1 = torch.sort(logits, dim=1, descending=True)
    _, o2 = torch.sort(o1, dim=1)
    ranks = (o2[torch.arange(6), label] + 1).numpy()
    tie = torch.tensor([[1.0, 3.0, 3.0, 2.0]])
    _, t1 = torch.sort(tie, dim=1, descending=True)
    _, t2 = torch.sort(t1, dim=1)
    h = torch.from_numpy(rng.standard_normal((4, 10, 16), dtype=np.float32)).requires_grad_(True)
    rel_idx = torch.tensor([[1, 5], [2, 6], [3, 7], [1, 8]])
    qh = torch.tensor([1, 1, 2, 3])
    ah = torch.tensor([4, 5, 6, 7])
    ar = torch.arange(4)
    sim = (F.relu(F.cosine_similarity(h[ar, qh], h[ar, ah])) + 1 - F.cosine_similarity(h[ar, rel_idx[ar, 0]], h[ar, rel_idx[ar, 1]])).mean(0)
    sim.backward()
    np.savez_compressed(os.path.join(out_dir, "g3_loss_rank.npz"), logits=logits.numpy(), label=label.numpy(),
                        lsce=np.float64(float(loss)), lsce_grad=lg.grad.numpy(), ranks=ranks,
                        tie_logits=tie.numpy(), tie_ranks_all=(t2 + 1).numpy(),
                        h=h.detach().numpy(), rel_idx=rel_idx.numpy(), q_head_idx=qh.numpy(), a_head_idx=ah.numpy(),
                        sim=np.float64(float(sim)), sim_grad=h.grad.numpy())
    print("G3 ok", float(loss), ranks, (t2 + 1).numpy())


def g3b(out_dir):
    """LabelSmoothSoftmaxCEV1 with ignored labels (lit_models/utils.py:47-62): six rows, two of them ignore_index = -100; reductions
    'mean' (sum / n_valid), 'sum' and per-row ('none'), each with the gradient of the logits."""
    from lit_models.utils import LabelSmoothSoftmaxCEV1
    rng = np.random.default_rng(33)
    logits = torch.from_numpy(rng.standard_normal((6, 37), dtype=np.float32) * 3)
    label = torch.from_numpy(rng.integers(0, 37, size=6))
    label[1] = -100
    label[4] = -100
    out = dict(logits=logits.numpy(), label=label.numpy(), ignore_index=np.int64(-100))
    w = torch.from_numpy(rng.standard_normal(6).astype(np.float32))
    for red in ("mean", "sum", "none"):
        lg = logits.clone().requires_grad_(True)
        loss = LabelSmoothSoftmaxCEV1(lb_smooth=0.1, reduction=red)(lg, label)
        (loss if red != "none" else (loss * w).sum()).backward()
        out["loss_" + red] = loss.detach().numpy().astype(np.float64)
        out["grad_" + red] = lg.grad.numpy()
    out["none_weights"] = w.numpy()
    # every row ignored: 0 / 0 (utils.py:60)
    allign = LabelSmoothSoftmaxCEV1(lb_smooth=0.1)(logits.clone(), torch.full((6,), -100, dtype=torch.long))
    out["loss_mean_all_ignored_isnan"] = np.bool_(bool(torch.isnan(allign)))
    np.savez_compressed(os.path.join(out_dir, "g3b_lsce_ignore.npz"), **out)
    print("G3b ok", out["loss_mean"], out["loss_sum"], out["loss_none"], bool(torch.isnan(allign)))


def load_reference_flava():
    """The reference's FLAVA module imported in place, with the extra shims of SURVEY 8(c): prune helpers stubbed into
    transformers.modeling_utils; get_head_mask / get_extended_attention_mask of FlavaPreTrainedModel overridden with the
    transformers==4.19.0 semantics the reference was written against."""
    import transformers.modeling_utils as mu
    if not hasattr(mu, "find_pruneable_heads_and_indices"):
        mu.find_pruneable_heads_and_indices = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())
    if not hasattr(mu, "prune_linear_layer"):
        mu.prune_linear_layer = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())
    spec = importlib.util.spec_from_file_location("ref_flava", os.path.join(REF, "models/modeling_flava.py"))
    fl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fl)
    fl.FlavaPreTrainedModel.get_head_mask = lambda self, head_mask, n, *a, **k: [None] * n
    fl.FlavaPreTrainedModel.get_extended_attention_mask = \
        lambda self, mask, shape=None, device=None, *a, **k: (1.0 - mask[:, None, None, :].to(torch.float32)) * -10000.0
    return fl


def g5_flava(out_dir):
    """G5: tiny FLAVA end-to-end (FlavaForMaskedLM as MarT runs it): trans_hidden, mask-row logits, fine-tune loss, gradients.
    Extra shims: prune helpers stubbed into transformers.modeling_utils; get_head_mask / get_extended_attention_mask
    of FlavaPreTrainedModel overridden with the transformers==4.19.0 semantics the reference was written against."""
    from oracle import flava_oracle as FO
    fl = load_reference_flava()
    from transformers import FlavaConfig
    V = TINY_BASE + TINY_E + TINY_R + 1
    c = FO.FlavaCfg(vocab_size=V, hidden_size=64, text_layers=3, image_layers=3, mm_layers=2, num_attention_heads=4,
                    intermediate_size=128, max_position_embeddings=64, image_size=64, patch_size=32)
    common = dict(hidden_size=64, num_attention_heads=4, intermediate_size=128)
    cfg = FlavaConfig(text_config=dict(vocab_size=V, num_hidden_layers=3, max_position_embeddings=64, **common),
                      image_config=dict(num_hidden_layers=3, image_size=64, patch_size=32, **common),
                      multimodal_config=dict(num_hidden_layers=2, **common), hidden_size=64, projection_dim=64)
    torch.manual_seed(0)
    model = fl.FlavaForMaskedLM(cfg)
    model.cls.decoder.weight = model.flava.text_model.embeddings.word_embeddings.weight      # tie manually (no resize under 5.x)
    # MarT always resizes the embedding (lit_models/transformer.py:33); transformers==4.19.0's _get_resized_embeddings builds a
    # plain nn.Embedding(new_num_tokens, dim) WITHOUT padding_idx, so the [PAD] row does receive gradient in real use
    model.flava.text_model.embeddings.word_embeddings.padding_idx = None
    sd = FO.init_params(c, seed=31)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(("position_ids" in m or "decoder" in m or "token_type_ids" in m) for m in missing), missing
    named = dict(model.named_parameters())
    assert set(named) == set(sd), (set(named) ^ set(sd))
    assert [n for n in named] == [n for n in sd], "parameter order differs"
    model.eval()
    batch = tiny_batch(B=3, L=24, seed=9)
    ids = batch["analogy_entity_ids"]
    out, trans = model(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], token_type_ids=batch["token_type_ids"],
                       pixel_values=batch["pixel_values"], sep_idx=batch["sep_idx"], return_dict=True)
    B = trans.shape[0]
    _, mask_idx = (batch["input_ids"] == 103).nonzero(as_tuple=True)
    mask_logits = out.logits[torch.arange(B), mask_idx][:, ids]
    loss = O.label_smooth_ce(mask_logits, batch["label"], 0.1) + 0.45 * O.relaxation_loss(trans, batch["rel_idx"], batch["q_head_idx"],
                                                                                          batch["a_head_idx"])
    loss.backward()
    none_grad = sorted(n for n, p in named.items() if p.grad is None)
    zero_grad = sorted(n for n, p in named.items() if p.grad is not None and float(p.grad.abs().max()) == 0.0)
    gnames = ["flava.text_model.encoder.layer.1.attention.attention.adaptive_weight.0",
              "flava.text_model.encoder.layer.1.attention.attention.adaptive_weight.1",
              "flava.text_model.encoder.layer.0.attention.attention.query.weight",
              "flava.text_model.embeddings.word_embeddings.weight", "flava.image_model.embeddings.position_embeddings",
              "flava.image_model.embeddings.cls_token", "flava.image_model.embeddings.patch_embeddings.projection.weight",
              "flava.image_model.embeddings.patch_embeddings.projection.bias",
              "flava.image_model.encoder.layer.2.output.dense.weight", "flava.multimodal_model.cls_token",
              "flava.multimodal_model.encoder.layer.1.attention.attention.key.bias", "flava.image_to_mm_projection.weight",
              "flava.text_to_mm_projection.bias", "flava.multimodal_model.layernorm.weight", "cls.transform.dense.weight", "cls.bias"]
    grads = {"grad::" + n: named[n].grad.detach().numpy().copy() for n in gnames}
    norms = {n: float(p.grad.norm()) for n, p in named.items() if p.grad is not None}
    np.savez_compressed(
        os.path.join(out_dir, "g5_flava_tiny.npz"),
        **{"in::" + k: v.numpy() for k, v in batch.items()}, weight_seed=np.int64(31),
        trans=trans.detach().numpy(), mask_logits=mask_logits.detach().numpy(), loss=np.float64(float(loss)),
        none_grad=np.array(none_grad), zero_grad=np.array(zero_grad),
        grad_norm_names=np.array(sorted(norms)), grad_norm_vals=np.array([norms[k] for k in sorted(norms)]), **grads)
    print("G5 flava loss", float(loss), "none-grad", len(none_grad), "zero-grad", len(zero_grad))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--only", default=None, help="g3b: only the ignored-label LSCE golden")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    torch.set_num_threads(8)
    lit, unimo = load_reference()
    g3b(a.out)
    if a.only == "g3b":
        return
    g3(lit, a.out)
    g2(unimo, a.out)
    g1_g4(lit, unimo, a.out)
    g5_flava(a.out)


if __name__ == "__main__":
    main()
