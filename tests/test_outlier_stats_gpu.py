"""Parity on weights with the activation statistics of TRAINED checkpoints ('nothing resembling trained CLIP / BERT statistics -- outlier
channels, large residual norms -- has ever gone through the bf16 path').  No checkpoint can be loaded offline; ``oracle.outlier_weights`` synthesises the two
properties: five outlier channels that every LayerNorm scales x12 and shifts by +2 (post-LN entries ~10-30, large per-row means) and residual projections scaled
x2 (f32 residual streams that grow from layer to layer), with the fusion softmax kept smooth (the reference's own bf16-weight control moves the logits by ~1e-2,
as on the conditioned set).  Checked against the fp32 CPU oracle at real dimensions:
  * the fp32-accurate path (evaluation default): logits within north_star's 1e-3, taken relative to the logit scale, ranks exact where the margin allows;
  * the bf16 training path: logits within 3 x the reference's own bf16-weight control (+ 2e-3) -- the same gate the plain-weight tests use; whether the absolute
    1e-2 of north_star is met is printed, not asserted (the control itself sits at ~1e-2 here); loss, gradient norms and directions."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mkgformer_oracle as O  # noqa: E402  (tests may use the oracle; the product never does)
from tests.test_model_gpu import BASE, NE, NR, _stats  # noqa: E402


def _build(patch, seed):
    import argparse
    from mkg_analogy_amd import data_synth as D
    from mkg_analogy_amd.lit_models import TransformerLitModel
    from mkg_analogy_amd.models import MKGformerKGC, TextConfig, VisionConfig
    torch.manual_seed(0)
    model = MKGformerKGC(VisionConfig(patch_size=patch), TextConfig())
    cfg = D.data_config(seed=1234)
    args = argparse.Namespace(label_smoothing=0.1, alpha=0.43, pretrain=0, lr=5e-5, weight_decay=0.01, optimizer="AdamW", warm_up_radio=0.1)
    lit = TransformerLitModel(model=model, args=args, tokenizer=D.FakeTokenizer(), data_config=cfg)
    vc = O.VisionCfg(patch_size=patch)
    sd0 = O.outlier_weights(O.init_params(vc, O.TextCfg(vocab_size=BASE + NE + NR), seed=seed))
    model.load_state_dict(sd0, strict=False)
    model.cuda()
    lit._init_relation_word()
    sd = O.init_relation_word({k: v.clone() for k, v in sd0.items()}, cfg["analogy_relation_ids"])
    return model, lit, cfg, vc, sd


@pytest.mark.parametrize("patch,B,L", [(32, 4, 64), (16, 2, 57)])
def test_outlier_channels_and_large_residuals(patch, B, L):
    from mkg_analogy_amd import data_synth as D
    model, lit, cfg, vc, sd = _build(patch, seed=7)
    tc = O.TextCfg(vocab_size=BASE + NE + NR + 1)
    batch = D.make_batch(B, L, seed=13)
    ids = torch.tensor(cfg["analogy_entity_ids"])
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    taps = {}
    _, trans_ref = O.forward(sdg, vc, tc, batch["input_ids"], batch["attention_mask"], batch["token_type_ids"], batch["pixel_values"], batch["sep_idx"], train=False)
    loss_ref, ml_ref = O.finetune_loss(sdg, trans_ref, batch["input_ids"], batch["label"], batch["rel_idx"], batch["q_head_idx"], batch["a_head_idx"], ids, alpha=0.43)
    loss_ref.backward()
    # the statistics this test is about, measured on the oracle's own streams: post-LayerNorm outliers and the size of the text residual stream
    # the reference's own bf16 sensitivity on this weight set (fp32 math, only the weight matrices rounded to bf16): logits AND gradients
    sdb = {k: (v.detach().to(torch.bfloat16).float() if v.dim() >= 2 and "embeddings" not in k else v.detach().clone()).requires_grad_(True) for k, v in sd.items()}
    _, trans_ctl = O.forward(sdb, vc, tc, batch["input_ids"], batch["attention_mask"], batch["token_type_ids"], batch["pixel_values"], batch["sep_idx"], train=False)
    loss_ctl, ml_ctl = O.finetune_loss(sdb, trans_ctl, batch["input_ids"], batch["label"], batch["rel_idx"], batch["q_head_idx"], batch["a_head_idx"], ids, alpha=0.43)
    loss_ctl.backward()
    ml_ctl = ml_ctl.detach()
    scale = max(1.0, float(ml_ref.detach().abs().max()))
    ctl = float((ml_ctl - ml_ref.detach()).abs().max())
    t = trans_ref.detach()
    print(f"\npatch {patch} B {B} L {L}: logit scale {scale:.2f}; trans_hidden |max| {float(t.abs().max()):.1f} row-mean/std max {float((t.mean(-1).abs() / t.std(-1)).max()):.2f}; "
          f"bf16-weight control of the reference: max|dlogit| {ctl:.3e}")
    gb = {k: v.cuda() for k, v in batch.items()}
    keys = ("input_ids", "attention_mask", "token_type_ids", "pixel_values", "sep_idx")
    model.eval()
    # ---- fp32-accurate path (the evaluation default)
    model.set_precision("fp32")
    with torch.no_grad():
        out, _ = model(**{k: gb[k] for k in keys}, return_dict=True)
        ml32 = out.logits.mask_rows(gb["input_ids"], 103)[:, ids.cuda()].float().cpu()
    model.set_precision("bf16")
    e32 = float((ml32 - ml_ref.detach()).abs().max())
    print(f"   fp32-accurate path: max|dlogit| {e32:.3e} (tolerance 1e-3 x scale = {1e-3 * scale:.3e})")
    assert e32 < 1e-3 * scale
    r32 = (ml32 > ml32[torch.arange(B), batch["label"]][:, None]).sum(1) + 1
    rref = O.ranks_double_sort(ml_ref.detach(), batch["label"])
    amb = ((ml_ref.detach() - ml_ref.detach()[torch.arange(B), batch["label"]][:, None]).abs() < 2 * e32).sum(1).numpy() - 1
    assert np.all(np.abs(r32.numpy() - rref) <= amb)
    # ---- bf16 training path
    st = model.store
    st.zero_grad()
    loss = lit.training_step(dict(gb), 1)
    loss.backward()
    torch.cuda.synchronize()
    with torch.no_grad():
        out, trans = model(**{k: gb[k] for k in keys}, return_dict=True)
        ml = out.logits.mask_rows(gb["input_ids"], 103)[:, ids.cuda()].float().cpu()
    e_t, r_t = _stats("trans_hidden (bf16 path)", trans, trans_ref)
    e_l = float((ml - ml_ref.detach()).abs().max())
    print(f"   bf16 path: max|dlogit| {e_l:.3e} (3 x control + 2e-3 = {3 * ctl + 2e-3:.3e}); loss hip {float(loss.detach()):.5f} oracle {float(loss_ref.detach()):.5f}")
    print(f"   north_star's absolute 1e-2 on logits: {'met' if e_l < 1e-2 else 'NOT met'} ({e_l:.3e}; the reference's own bf16-weight control: {ctl:.3e})")
    assert e_l < 3.0 * ctl + 2e-3
    assert r_t < 2e-2
    assert abs(float(loss) - float(loss_ref)) < 3.0 * ctl + 5e-3
    assert bool(torch.isfinite(st.grad).all())
    worst = 0.0
    for n in ("cls.predictions.transform.dense.weight", "unimo.text_embeddings.LayerNorm.weight", "unimo.vision_pre_layrnorm.weight",
              "unimo.encoder.text_layer.11.output.LayerNorm.weight", "unimo.encoder.text_layer.8.intermediate.fusion_dense.weight",
              "unimo.encoder.text_layer.5.attention.self.query.weight", "unimo.encoder.text_layer.0.output.dense.weight",
              "unimo.encoder.vision_layers.11.mlp.fc2.weight", "unimo.encoder.vision_layers.8.self_attn.q_proj.weight",
              "unimo.encoder.vision_layers.5.layer_norm1.weight", "unimo.encoder.vision_layers.0.mlp.fc1.weight",
              "unimo.vision_embeddings.patch_embedding.weight", "unimo.text_embeddings.word_embeddings.weight"):
        g, r, c = st.g(n).detach().float().cpu(), sdg[n].grad, sdb[n].grad
        rel = float((g - r).norm() / (r.norm() + 1e-20))
        crel = float((c - r).norm() / (r.norm() + 1e-20))
        cos = float(torch.nn.functional.cosine_similarity(g.flatten(), r.flatten(), dim=0))
        worst = max(worst, rel)
        print(f"   grad {n}: rel-L2 {rel:.3e} (reference's bf16-weight control {crel:.3e}) cos {cos:.5f} |ref| {float(r.norm()):.3e}")
        assert rel < 3.0 * crel + 0.05, n                        # within 3 x the displacement the reference's own math shows under bf16 weights (+ 5 %)
