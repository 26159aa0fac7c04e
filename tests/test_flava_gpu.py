"""FLAVA backbone (SURVEY 8(a) row 19 / BASELINE config 4) on the HIP path vs the CPU oracle, real dimensions (768 wide,
12 + 12 + 6 layers, 393 image tokens, L = 64, vocabulary 42007)."""
import argparse

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import flava_oracle as FO  # noqa: E402
from oracle import mkgformer_oracle as O  # noqa: E402


@pytest.mark.parametrize("B,L", [(2, 64), (3, 37)])        # (3, 37): odd batch, 431 multimodal tokens (no multiple of 8), general attention paths
def test_flava_forward_backward_vs_oracle(B, L):
    from mkg_analogy_amd import data_synth as D
    from mkg_analogy_amd.lit_models import TransformerLitModel
    from mkg_analogy_amd.models import FlavaKGC, flava_config
    c = FO.FlavaCfg(vocab_size=D.VOCAB - 1)
    torch.manual_seed(0)
    model = FlavaKGC(flava_config(vocab_size=30522))
    cfg = D.data_config(seed=1234)
    args = argparse.Namespace(label_smoothing=0.1, alpha=0.45, pretrain=0, lr=5e-5, weight_decay=0.01, optimizer="AdamW", warm_up_radio=0.1)
    lit = TransformerLitModel(model=model, args=args, tokenizer=D.FakeTokenizer(), data_config=cfg)      # resize -> 42006
    sd0 = FO.init_params(c, seed=13)
    missing, unexpected = model.load_state_dict(sd0, strict=False)
    assert not unexpected and all(("position_ids" in m or "decoder" in m) for m in missing), (missing, unexpected)
    model.cuda()
    lit._init_relation_word()                                                                            # -> 42007
    W = sd0["flava.text_model.embeddings.word_embeddings.weight"]
    sd = dict(sd0)
    sd["flava.text_model.embeddings.word_embeddings.weight"] = torch.cat([W, W[torch.tensor(cfg["analogy_relation_ids"])].mean(0, keepdim=True)], 0)
    sd["cls.bias"] = torch.cat([sd0["cls.bias"], torch.zeros(1)])
    c = FO.FlavaCfg(vocab_size=D.VOCAB)
    batch = D.make_batch(B, L, seed=17)
    ids = torch.tensor(cfg["analogy_entity_ids"])
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    trans_ref = FO.forward(sdg, c, batch["input_ids"], batch["attention_mask"], batch["token_type_ids"], batch["pixel_values"], batch["sep_idx"])
    _, mi = (batch["input_ids"] == 103).nonzero(as_tuple=True)
    ml_ref = FO.score(sdg, trans_ref[torch.arange(B), mi], ids)
    loss_ref = O.label_smooth_ce(ml_ref, batch["label"], 0.1) + 0.45 * O.relaxation_loss(trans_ref, batch["rel_idx"], batch["q_head_idx"],
                                                                                         batch["a_head_idx"])
    loss_ref.backward()
    model.eval()
    gb = {k: v.cuda() for k, v in batch.items()}
    st = model.store
    st.zero_grad()
    loss = lit.training_step(dict(gb), 1)
    loss.backward()
    torch.cuda.synchronize()
    out, trans = model(input_ids=gb["input_ids"], attention_mask=gb["attention_mask"], token_type_ids=gb["token_type_ids"],
                       pixel_values=gb["pixel_values"], sep_idx=gb["sep_idx"], return_dict=True)
    ml = out.logits[torch.arange(B, device="cuda"), mi.cuda()][:, ids.cuda()]
    rel_t = float((trans.detach().float().cpu() - trans_ref.detach()).norm() / trans_ref.detach().norm())
    e_l = float((ml.detach().float().cpu() - ml_ref.detach()).abs().max())
    scale = max(1.0, float(ml_ref.detach().abs().max()))
    print(f"\nflava: loss hip {float(loss.detach()):.5f} oracle {float(loss_ref.detach()):.5f}; trans rel-L2 {rel_t:.3e}; logits max|err| {e_l:.3e} (scale {scale:.2f})")
    assert rel_t < 4e-3 and e_l < 4e-3 and abs(float(loss.detach()) - float(loss_ref.detach())) < 5e-3       # measured 7.5e-4 / 1.6e-3 / 7e-4
    ev = lit._eval(dict(gb), 0)
    ranks_ref = O.ranks_double_sort(ml_ref.detach(), batch["label"])
    amb = ((ml_ref.detach() - ml_ref.detach()[torch.arange(B), batch["label"]][:, None]).abs() < 2 * e_l).sum(1).numpy() - 1
    assert np.all(np.abs(ev["entity_ranks"] - ranks_ref) <= amb)
    names = ["cls.transform.dense.weight", "cls.bias", "flava.text_model.embeddings.word_embeddings.weight",
             "flava.image_model.embeddings.position_embeddings", "flava.image_model.embeddings.cls_token",
             "flava.image_model.embeddings.patch_embeddings.projection.weight", "flava.image_model.embeddings.patch_embeddings.projection.bias",
             "flava.multimodal_model.cls_token", "flava.image_to_mm_projection.weight", "flava.text_to_mm_projection.bias",
             "flava.multimodal_model.layernorm.weight", "flava.text_model.embeddings.LayerNorm.bias"]
    for mod, ls in (("text_model", (0, 6, 11)), ("image_model", (0, 11)), ("multimodal_model", (0, 5))):
        for l in ls:
            p = f"flava.{mod}.encoder.layer.{l}."
            names += [p + "attention.attention.query.weight", p + "attention.attention.key.bias", p + "attention.attention.value.weight",
                      p + "attention.output.dense.weight", p + "layernorm_before.weight", p + "intermediate.dense.weight",
                      p + "output.dense.bias", p + "layernorm_after.bias"]
            if mod == "text_model":
                names += [p + "attention.attention.adaptive_weight.0", p + "attention.attention.adaptive_weight.1"]
    for n in names:
        g, r = st.g(n).detach().float().cpu().reshape(-1), sdg[n].grad.reshape(-1)
        if r.norm().item() < 1e-7:
            assert g.norm().item() < 1e-3, n
            continue
        rel = ((g - r).norm() / r.norm()).item()
        cos = torch.nn.functional.cosine_similarity(g, r, dim=0).item()
        print(f"   grad {n}: rel-L2 {rel:.3e} cos {cos:.5f} |ref| {r.norm().item():.3e} |got| {g.norm().item():.3e}")
        if g.numel() == 1:      # scalar = sum of signed per-score terms: compare on the scale of the other layers' values
            assert abs(float(g) - float(r)) < 0.12 * abs(float(r)) + 3e-3, n
        else:
            assert cos > 0.99 and rel < 0.12, n
    for n in ("flava.logit_scale", "flava.image_model.pooler.dense.weight", "flava.text_model.layernorm.weight",
              "flava.image_model.encoder.layer.3.attention.attention.adaptive_weight.0"):
        assert float(st.g(n).abs().max()) == 0.0
    # ---- fp32-accurate evaluation path (engine_precise.PreciseFlavaForward): north_star's 1e-3 gate on logits
    model.eval()
    model.set_precision("fp32")
    out32, trans32 = model(input_ids=gb["input_ids"], attention_mask=gb["attention_mask"], token_type_ids=gb["token_type_ids"],
                           pixel_values=gb["pixel_values"], sep_idx=gb["sep_idx"], return_dict=True)
    ml32 = out32.logits[torch.arange(B, device="cuda"), mi.cuda()][:, ids.cuda()]
    e32 = float((ml32.cpu() - ml_ref.detach()).abs().max())
    t32 = float((trans32.cpu() - trans_ref.detach()).abs().max())
    print(f"   fp32 path: logits max|err| {e32:.3e}  trans max|err| {t32:.3e}")
    assert e32 < 1e-3 and t32 < 5e-3
    ev32 = lit._eval(dict(gb), 0)
    lab = ml_ref.detach()[torch.arange(B), batch["label"]]
    gap = (ml_ref.detach() - lab[:, None]).abs()
    gap[torch.arange(B), batch["label"]] = 1e9
    safe = (gap.min(1).values > 2 * e32).numpy()
    assert np.array_equal(ev32["entity_ranks"][safe], np.asarray(O.ranks_count(ml_ref.detach(), batch["label"]))[safe])
    model.set_precision("bf16")


def test_flava_vs_reference_at_real_dimensions():
    """G9 (oracle/gen_goldens_full.py:g9_flava): the UNMODIFIED reference FlavaForMaskedLM at real dimensions (B=2, 393 image
    tokens, L=64, V=42007) against the HIP path directly -- no oracle in between."""
    import os
    from mkg_analogy_amd import data_synth as D
    from mkg_analogy_amd.lit_models import TransformerLitModel
    from mkg_analogy_amd.models import FlavaKGC, flava_config
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g9_flava_real.npz"), allow_pickle=False))
    c = FO.FlavaCfg(vocab_size=D.VOCAB - 1)
    torch.manual_seed(0)
    model = FlavaKGC(flava_config(vocab_size=30522))
    cfg = D.data_config(seed=1234)
    args = argparse.Namespace(label_smoothing=0.1, alpha=0.45, pretrain=0, lr=5e-5, weight_decay=0.01, optimizer="AdamW", warm_up_radio=0.1)
    lit = TransformerLitModel(model=model, args=args, tokenizer=D.FakeTokenizer(), data_config=cfg)      # resize -> 42006
    sd0 = FO.init_params(c, seed=int(g["weight_seed"]))
    missing, unexpected = model.load_state_dict(sd0, strict=False)
    assert not unexpected and all(("position_ids" in m or "decoder" in m) for m in missing), (missing, unexpected)
    model.cuda()
    lit._init_relation_word()                                                                            # -> 42007, [R] = mean of the relation rows
    B = int(g["B"])
    batch = D.make_batch(B, int(g["L"]), seed=int(g["batch_seed"]))
    for k, v in batch.items():
        if k != "pixel_values":
            assert np.array_equal(v.numpy(), g["in::" + k]), k
    assert abs(float(batch["pixel_values"].double().sum()) - float(g["pixel_sum"])) < 1e-6 * float(g["pixel_abs_sum"])
    gb = {k: v.cuda() for k, v in batch.items()}
    ids = torch.tensor(cfg["analogy_entity_ids"], device="cuda")
    ar = torch.arange(B, device="cuda")
    rows = torch.from_numpy(g["trans_row_index"]).cuda()
    ref_l, ref_t = torch.from_numpy(g["mask_logits"]), torch.from_numpy(g["trans_rows"])
    model.eval()
    keys = ("input_ids", "attention_mask", "token_type_ids", "pixel_values", "sep_idx")

    def forward():
        with torch.no_grad():
            out, trans = model(**{k: gb[k] for k in keys}, return_dict=True)
            return out.logits[ar, rows[:, 0]][:, ids].float().cpu(), trans[ar[:, None], rows].float().cpu()

    model.set_precision("fp32")
    ml32, tr32 = forward()
    ev32 = lit._eval(dict(gb), 0)
    model.set_precision("bf16")
    e32 = float((ml32 - ref_l).abs().max())
    print(f"\ng9 flava: fp32-accurate path max|dlogit| {e32:.3e}  trans rows max|err| {float((tr32 - ref_t).abs().max()):.3e}")
    assert e32 < 1e-3
    lab = ref_l[torch.arange(B), batch["label"]]
    near = ((ref_l - lab[:, None]).abs() < 2 * e32).sum(1).numpy() - 1
    assert np.all(np.abs(ev32["entity_ranks"] - g["ranks"]) <= near), (ev32["entity_ranks"], g["ranks"])
    ml, tr = forward()
    e_l, rms = float((ml - ref_l).abs().max()), float((ml - ref_l).pow(2).mean().sqrt())
    scale = max(1.0, float(ref_l.abs().max()))
    r_t = float((tr - ref_t).norm() / ref_t.norm())
    print(f"   bf16 path max|dlogit| {e_l:.3e} rms {rms:.3e} (logit scale {scale:.2f}) trans rows rel-L2 {r_t:.3e}   [text + multimodal stacks on fp16 operands: {model.engine.f16}]")
    f16_0, f16i_0 = model.engine.f16, model.engine.f16_img
    model.engine.f16 = model.engine.f16_img = False
    mlp, _ = forward()
    model.engine.f16, model.engine.f16_img = f16_0, f16i_0
    print(f"   MART_TEXT_F16=0 (all three stacks bf16): max|dlogit| {float((mlp - ref_l).abs().max()):.3e} rms {float((mlp - ref_l).pow(2).mean().sqrt()):.3e}")
    # north_star asks for 1e-2; round 5 (fp16 operands in all three stacks, patch embedding and the two multimodal projections on fp16 operands with f32
    # results: the projections read UN-normalised hidden states, whose bf16 copies were 3/4 of the error) measures 1.6e-3 -- gated at 4e-3
    assert r_t < 4e-3
    assert e_l < 4e-3, f"bf16-path logits vs the reference (FLAVA, forward products on fp16 operands): {e_l:.3e}"
    st = model.store
    st.zero_grad()
    loss = lit.training_step(dict(gb), 1)
    loss.backward()
    torch.cuda.synchronize()
    print(f"   loss hip {float(loss.detach()):.5f} reference {float(g['loss']):.5f}")
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-2
    norms = dict(zip(g["grad_norm_names"].tolist(), g["grad_norm_vals"].tolist()))
    worst = 0.0
    for n, ref in norms.items():
        if n.endswith("decoder.weight") or "adaptive_weight" in n or n not in st.slots:
            continue
        got = float(st.g(n).double().norm())
        if ref < 1e-7:
            assert got < 1e-3, (n, got, ref)
            continue
        worst = max(worst, abs(got - ref) / ref)
        assert abs(got - ref) / ref < 0.02, (n, got, ref)          # measured 0.66 % (round 4 gate: 10 %)
    bad = []
    for k in g:
        if not k.startswith("gs::"):
            continue
        n, ref = k[4:], g[k]
        f = st.g(n).detach().reshape(-1)
        step = max(1, f.numel() // 1024)
        got = f[::step][:1024].float().cpu().numpy()
        if np.linalg.norm(ref) < 1e-7:
            continue
        if ref.size == 1:
            if abs(float(got[0]) - float(ref[0])) > 0.12 * abs(float(ref[0])) + 3e-3:
                bad.append((n, float(got[0]), float(ref[0])))
            continue
        rel = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
        cos = float(got @ ref / (np.linalg.norm(got) * np.linalg.norm(ref)))
        if rel > 0.12 or cos < 0.99:
            bad.append((n, rel, cos))
    print(f"   gradients: worst |norm| deviation {worst:.3e} over {len(norms)} tensors; {sum(k.startswith('gs::') for k in g)} sampled tensors")
    assert not bad, bad


def _g9b_setup():
    import os
    from mkg_analogy_amd import data_synth as D
    from mkg_analogy_amd.lit_models import TransformerLitModel
    from mkg_analogy_amd.models import FlavaKGC, flava_config
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g9b_flava_b8.npz"), allow_pickle=False))
    c = FO.FlavaCfg(vocab_size=D.VOCAB - 1)
    torch.manual_seed(0)
    model = FlavaKGC(flava_config(vocab_size=30522))
    cfg = D.data_config(seed=1234)
    args = argparse.Namespace(label_smoothing=0.1, alpha=0.45, pretrain=0, lr=5e-5, weight_decay=0.01, optimizer="AdamW", warm_up_radio=0.1)
    lit = TransformerLitModel(model=model, args=args, tokenizer=D.FakeTokenizer(), data_config=cfg)
    missing, unexpected = model.load_state_dict(FO.init_params(c, seed=int(g["weight_seed"])), strict=False)
    assert not unexpected and all(("position_ids" in m or "decoder" in m) for m in missing), (missing, unexpected)
    model.cuda()
    lit._init_relation_word()
    B = int(g["B"])
    batch = D.make_batch(B, int(g["L"]), seed=int(g["batch_seed"]))
    for k, v in batch.items():
        if k != "pixel_values":
            assert np.array_equal(v.numpy(), g["in::" + k]), k
    assert abs(float(batch["pixel_values"].double().sum()) - float(g["pixel_sum"])) < 1e-6 * float(g["pixel_abs_sum"])
    return g, model, lit, cfg, batch


def _tap_errors(g, taps):
    """rel-L2 of every per-layer tap against the reference's (the token rows the golden keeps, oracle/gen_goldens_full.py:g9b_tap_rows)."""
    out = {}
    for k in sorted(x[5:] for x in g if x.startswith("tap::")):
        ref = g["tap::" + k]
        S = taps[k].shape[1]
        rows = np.unique(np.concatenate([np.arange(0, S, 16), np.array([0, 1, S - 1])]))
        got = taps[k][:, torch.from_numpy(rows).to(taps[k].device)].float().cpu().numpy()
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        out[k] = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
    return out


def test_flava_b8_per_layer_vs_reference():
    """G9b: the UNMODIFIED reference FlavaForMaskedLM at real dimensions, B = 8, with per-layer taps (text 0 / 6 / 11, image 0 / 11, multimodal 0 / 5).
    bf16 path (text + multimodal stacks on fp16 operands): every tap at rounding level, logits within north_star's 1e-2, loss, ranks, gradient norms."""
    g, model, lit, cfg, batch = _g9b_setup()
    B = int(g["B"])
    gb = {k: v.cuda() for k, v in batch.items()}
    ids = torch.tensor(cfg["analogy_entity_ids"], device="cuda")
    ar = torch.arange(B, device="cuda")
    rows = torch.from_numpy(g["trans_row_index"]).cuda()
    ref_l = torch.from_numpy(g["mask_logits"])
    model.eval()
    keys = ("input_ids", "attention_mask", "token_type_ids", "pixel_values", "sep_idx")
    model.engine.taps = {}
    with torch.no_grad():
        out, trans = model(**{k: gb[k] for k in keys}, return_dict=True)
        ml = out.logits[ar, rows[:, 0]][:, ids].float().cpu()
    errs = _tap_errors(g, model.engine.taps)
    model.engine.taps = None
    e_l, rms = float((ml - ref_l).abs().max()), float((ml - ref_l).pow(2).mean().sqrt())
    print(f"\ng9b flava B=8: bf16 path taps rel-L2 { {k: round(v, 5) for k, v in errs.items()} }\n   logits max|dlogit| {e_l:.3e} rms {rms:.3e}")
    assert all(v < 3e-3 for v in errs.values()), errs             # measured 2e-4 ... 8e-4 at every tap of the three stacks (round 4: up to 3.3e-3)
    assert e_l < 4e-3, f"bf16-path logits vs the reference (measured 1.9e-3; north_star 1e-2): {e_l:.3e}"
    st = model.store
    st.zero_grad()
    loss = lit.training_step(dict(gb), 1)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-2
    ev = lit._eval(dict(gb), 0)
    lab = ref_l[torch.arange(B), batch["label"]]
    amb = ((ref_l - lab[:, None]).abs() < 2 * e_l).sum(1).numpy() - 1
    assert np.all(np.abs(ev["entity_ranks"] - g["ranks"]) <= amb), (ev["entity_ranks"], g["ranks"], amb)
    norms = dict(zip(g["grad_norm_names"].tolist(), g["grad_norm_vals"].tolist()))
    worst = 0.0
    for n, ref in norms.items():
        if n.endswith("decoder.weight") or "adaptive_weight" in n or n not in st.slots or ref < 1e-7:
            continue
        worst = max(worst, abs(float(st.g(n).double().norm()) - ref) / ref)
    print(f"   loss hip {float(loss.detach()):.5f} reference {float(g['loss']):.5f}; worst gradient-norm deviation {worst:.3e}")
    assert worst < 0.02                                           # measured 0.57 % (round 4 gate: 5 %)


def test_flava_fp32_training_step_vs_reference():
    """The fp32-accurate FLAVA training step (engine_precise.PreciseFlavaTrain: set_precision("fp32") with gradients enabled) against the reference's
    own fp32 step on G9b: per-layer taps, loss, and ALL gradient norms / sampled tensors to 1e-3 (what test_fp32_training_step_vs_reference holds
    MKGformer to)."""
    g, model, lit, cfg, batch = _g9b_setup()
    gb = {k: v.cuda() for k, v in batch.items()}
    model.eval()
    model.set_precision("fp32")
    st = model.store
    st.zero_grad()
    # the engine is created by the first forward call: run the step once to build it, then again with taps
    loss = lit.training_step(dict(gb), 1)
    loss.backward()
    st.zero_grad()
    model._precise_train.taps = {}
    loss = lit.training_step(dict(gb), 1)
    loss.backward()
    torch.cuda.synchronize()
    errs = _tap_errors(g, model._precise_train.taps)
    model._precise_train.taps = None
    model.set_precision("bf16")
    dl = abs(float(loss.detach()) - float(g["loss"]))
    norms = dict(zip(g["grad_norm_names"].tolist(), g["grad_norm_vals"].tolist()))
    worst, wn, aw_ref, aw_got = 0.0, "", [], []
    for n, ref in norms.items():
        if n.endswith("decoder.weight") or n not in st.slots:
            continue
        got = float(st.g(n).double().norm())
        if ref < 1e-7:
            assert got < 1e-5, (n, got, ref)
            continue
        if "adaptive_weight" in n:
            aw_ref.append(ref); aw_got.append(got)
            continue
        r = abs(got - ref) / ref
        if r > worst:
            worst, wn = r, n
    r_aw = float(np.linalg.norm(np.array(aw_got) - np.array(aw_ref)) / np.linalg.norm(aw_ref)) if aw_ref else 0.0
    rels = []
    for k in g:
        if k.startswith("gs::") and np.linalg.norm(g[k]) >= 1e-7 and "adaptive_weight" not in k:
            f = st.g(k[4:]).detach().reshape(-1)
            step = max(1, f.numel() // 1024)
            got = f[::step][:1024].float().cpu().numpy()
            rels.append((float(np.linalg.norm(got - g[k]) / np.linalg.norm(g[k])), k[4:]))
    rels.sort(reverse=True)
    print(f"\ng9b flava fp32 training step: loss hip {float(loss.detach()):.7f} reference {float(g['loss']):.7f}; taps rel-L2 max {max(errs.values()):.2e}; "
          f"worst gradient-norm deviation {worst:.3e} ({wn}); adaptive-weight gradients as a vector rel-L2 {r_aw:.3e}; worst sample rel-L2 {rels[0][0]:.3e} ({rels[0][1]})")
    for n in g["none_grad"].tolist():
        if n in st.slots and not n.endswith("decoder.weight"):
            assert float(st.g(n).abs().max()) == 0.0, n
    assert max(errs.values()) < 1e-4 and dl < 1e-4
    assert worst < 1e-3 and rels[0][0] < 1e-3 and r_aw < 1e-3, (worst, wn, rels[:3], r_aw)
