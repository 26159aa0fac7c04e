"""Parity at the benchmark's own shapes, against goldens written by the UNMODIFIED reference (oracle/gen_goldens_full.py):

  G7  BASELINE configs[1] shape (first 32 examples of bench.py's rank-0 batch, P=196, L=64, V=42007), plain N(0,0.02) weights --
      the network bench.py times -- and the well-conditioned variant; compared DIRECTLY with the HIP path (no oracle run on the
      GPU box): mask logits, trans_hidden rows, loss, ranks, every gradient norm and strided gradient samples.
  G8  BASELINE configs[4] shape (MarKG pre-train step, L=96, no sep_idx, mixed pre_type, full E=11292 / R=192 heads); G8b = the
      same step at the ViT-B/16 geometry (P=196, 393 vision tokens).
  +   teacher-forced per-layer parity on PLAIN weights: every layer is fed the oracle's inputs and upstream gradients, so the
      reported error is that layer's own bf16 rounding and not the chaos of the unscaled fusion softmax downstream.

Tolerances are ABSOLUTE on logits (north_star: 1e-3 fp32 / 1e-2 bf16) wherever the network is well conditioned: the DEFAULT
training / evaluation configuration (vision stream bf16, text stream's forward products on fp16 operands, split-precision head) is
held to max|dlogit| < 1e-2 against the reference on every conditioned golden (G7, G8, G8b) and on all 256 examples of the bench batch;
the fp32-accurate path to 1e-3, plain weights included.  Plain N(0,0.02) weights are chaotic in the unscaled fusion softmax: bf16
bounds there are multiples of the reference's own bf16-weight control, and the per-layer teacher-forced test carries the parity claim.
"""
import argparse
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mkgformer_oracle as O  # noqa: E402  (tests may use the oracle; the product never does)

BASE, NE, NR = 30522, 11292, 192
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(tag):
    return dict(np.load(os.path.join(GOLD, tag + ".npz"), allow_pickle=False))


def _product(g, pretrain=False):
    """The HIP model with the golden's weights (regenerated from its numpy seed) and the trainer surface around it."""
    from mkg_analogy_amd import data_synth as D
    from mkg_analogy_amd.lit_models import TransformerLitModel
    from mkg_analogy_amd.models import MKGformerKGC, TextConfig, VisionConfig
    patch = int(g["patch"])
    torch.manual_seed(0)
    model = MKGformerKGC(VisionConfig(patch_size=patch), TextConfig())
    cfg = D.data_config(seed=1234)
    args = argparse.Namespace(label_smoothing=0.1, alpha=0.43, pretrain=int(pretrain), lr=5e-5, weight_decay=0.01, optimizer="AdamW",
                              warm_up_radio=0.1)
    lit = TransformerLitModel(model=model, args=args, tokenizer=D.FakeTokenizer(), data_config=cfg)
    vc = O.VisionCfg(patch_size=patch)
    sd0 = O.init_params(vc, O.TextCfg(vocab_size=BASE + NE + NR), seed=int(g["weight_seed"]))
    if int(g["conditioned"]):
        sd0 = O.condition_weights(sd0)
    missing, unexpected = model.load_state_dict(sd0, strict=False)
    assert not unexpected and all(("position_ids" in m or "decoder" in m) for m in missing), (missing, unexpected)
    model.cuda()
    lit._init_relation_word()
    return model, lit, cfg


def _batch(g, pretrain=False):
    """The golden's batch, regenerated from its seed; the integer inputs and the pixel checksum are checked against the file."""
    from mkg_analogy_amd import data_synth as D
    B = int(g["B"])
    full = D.make_batch(int(g["batch_total"]), int(g["L"]), seed=int(g["batch_seed"]), pretrain=pretrain)
    b = {k: v[:B].clone() for k, v in full.items()}
    for k, v in b.items():
        if k != "pixel_values":
            assert np.array_equal(v.numpy(), g["in::" + k]), k
    assert abs(float(b["pixel_values"].double().sum()) - float(g["pixel_sum"])) < 1e-6 * float(g["pixel_abs_sum"])
    return b


def _sample(t, n=1024):
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].float().cpu().numpy()


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def _cos(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))


def _grad_report(st, g, tol_rel, tol_cos, tol_norm, skip=(), ctl_mult=0.0):
    """Every gradient norm and the strided samples of ~45 tensors against the reference's.  With ``ctl_mult`` the bound of a
    tensor is tol + ctl_mult x the displacement of the same quantity in the golden's sensitivity control (the reference itself
    with bf16-rounded weight matrices): the network's own bf16 sensitivity, measured per tensor."""
    norms = dict(zip(g["grad_norm_names"].tolist(), g["grad_norm_vals"].tolist()))
    cnorms = dict(zip(g["grad_norm_names"].tolist(), g["ctl::grad_norm_vals"].tolist())) if ctl_mult else {}
    worst_n, worst_s, bad = 0.0, 0.0, []
    # the adaptive-weight gradients are SCALARS -- sums of signed score x d(score) terms with heavy cancellation, a few 1e-5 in
    # size -- so they are compared as ONE vector (the sampled layers' eight scalars), not one by one
    aw = sorted(k for k in g if k.startswith("gs::") and "adaptive_weight" in k)
    if aw:
        ref = np.concatenate([g[k].ravel() for k in aw])
        got = np.concatenate([_sample(st.g(k[4:])).ravel() for k in aw])
        r = _rel(got, ref)
        rc = _rel(np.concatenate([g["ctl::" + k].ravel() for k in aw]), ref) if ctl_mult else 0.0
        print(f"   adaptive-weight gradients (vector of {len(ref)}): rel-L2 {r:.3e}" + (f"   (control {rc:.3e})" if ctl_mult else ""))
        if r > 0.15 + ctl_mult * rc:
            bad.append(("adaptive", r, rc))
    skip = tuple(skip) + tuple(n for n in norms if "adaptive_weight" in n)
    med_cn = float(np.median([abs(cnorms[n] - r) / r for n, r in norms.items() if r >= 1e-7])) if ctl_mult else 0.0
    for n, ref in norms.items():
        if n.endswith("decoder.weight") or n in skip:
            continue
        got = float(st.g(n).double().norm())
        if ref < 1e-7:
            assert got < 1e-3, (n, got, ref)
            continue
        r = abs(got - ref) / ref
        lim = tol_norm + (ctl_mult * max(abs(cnorms[n] - ref) / ref, med_cn) if ctl_mult else 0.0)   # own or median control displacement
        worst_n = max(worst_n, r)
        if r > lim:
            bad.append(("norm", n, got, ref, lim))
    keys = [k for k in g if k.startswith("gs::") and np.linalg.norm(g[k]) >= 1e-7 and k[4:] not in skip]
    # the control displaces different tensors by very different amounts (chaotic regime: 0.04 ... 2.0 on the same batch), so a
    # tensor is held to the larger of its own and the median control displacement
    med_c = float(np.median([_rel(g["ctl::" + k], g[k]) for k in keys])) if ctl_mult else 0.0
    rels = []
    for k in keys:
        n = k[4:]
        ref = g[k]
        got = _sample(st.g(n))
        r, c = _rel(got, ref), _cos(got, ref)
        rc = max(_rel(g["ctl::" + k], ref), med_c) if ctl_mult else 0.0
        worst_s = max(worst_s, r)
        rels.append(r)
        print(f"   grad sample {n}: rel-L2 {r:.3e} cos {c:.5f}" + (f"   (control {_rel(g['ctl::' + k], ref):.3e})" if ctl_mult else ""))
        if r > tol_rel + ctl_mult * rc or (not ctl_mult and c < tol_cos):
            bad.append(("sample", n, r, c, rc))
    if ctl_mult:
        print(f"   median gradient displacement: HIP bf16 path {float(np.median(rels)):.3e}, control (reference, bf16-rounded weights) {med_c:.3e}")
        # as a whole in the range of the reference's own one-rounding control.  In the chaotic regime (plain weights: control ~1.9, i.e. the
        # reference's gradients are uncorrelated with its own bf16-weight rerun) the median of a run is a heavy-tailed sample -- 0.55 in one
        # kernel revision, 3.2 in the next, both with every per-layer teacher-forced bound green -- hence a factor, not an equality
        if float(np.median(rels)) > 2.0 * med_c + tol_rel:
            bad.append(("median", float(np.median(rels)), med_c))
    for n in g["none_grad"].tolist():                       # tensors the reference never touches stay at zero
        if n in st.slots and not n.endswith("decoder.weight"):
            assert float(st.g(n).abs().max()) == 0.0, n
    print(f"   gradients: worst |norm| deviation {worst_n:.3e} over {len(norms)} tensors, worst sample rel-L2 {worst_s:.3e}")
    assert not bad, bad
    return worst_n, worst_s


@pytest.mark.parametrize("tag", ["g7_bench_cond", "g7_bench_plain"])
def test_finetune_step_vs_reference_at_bench_shape(tag):
    g = _load(tag)
    cond = bool(int(g["conditioned"]))
    model, lit, cfg = _product(g)
    batch = _batch(g)
    B = int(g["B"])
    gb = {k: v.cuda() for k, v in batch.items()}
    ids = torch.tensor(cfg["analogy_entity_ids"], device="cuda")
    ar = torch.arange(B, device="cuda")
    rows = torch.from_numpy(g["trans_row_index"]).cuda()
    ref_logits, ref_trans = torch.from_numpy(g["mask_logits"]), torch.from_numpy(g["trans_rows"])
    model.eval()
    keys = ("input_ids", "attention_mask", "token_type_ids", "pixel_values", "sep_idx")

    def forward():
        with torch.no_grad():
            out, trans = model(**{k: gb[k] for k in keys}, return_dict=True)
            ml = out.logits[ar, rows[:, 0]][:, ids].float().cpu()
            return ml, trans[ar[:, None], rows].float().cpu()

    # ---- fp32-accurate evaluation path: north_star's 1e-3 absolute on logits, bit-exact ranked indices
    model.set_precision("fp32")
    ml32, tr32 = forward()
    ev32 = lit._eval(dict(gb), 0)
    model.set_precision("bf16")
    e32 = float((ml32 - ref_logits).abs().max())
    print(f"\n{tag}: fp32-accurate path  max|dlogit| {e32:.3e}  trans rows max|err| {float((tr32 - ref_trans).abs().max()):.3e}")
    assert e32 < 1e-3
    ref_ranks = g["ranks::entity_ranks"]
    lab = ref_logits[torch.arange(B), batch["label"]]
    margin = (ref_logits - lab[:, None]).abs()
    margin[torch.arange(B), batch["label"]] = 1e9
    safe = (margin.min(1).values > 2 * e32).numpy()          # rows whose rank cannot flip inside the measured logit error
    print(f"   ranks fp32 path {ev32['entity_ranks'].tolist()}\n   reference       {ref_ranks.tolist()}  ({int(safe.sum())}/{B} rows outside the error margin)")
    assert np.array_equal(ev32["entity_ranks"][safe], ref_ranks[safe])          # bit-exact wherever the margin exceeds the logit error
    near = ((ref_logits - lab[:, None]).abs() < 2 * e32).sum(1).numpy() - 1     # competitors inside the error band of the others
    assert np.all(np.abs(ev32["entity_ranks"] - ref_ranks) <= near)
    print(f"   fp32 path: {int((ev32['entity_ranks'] == ref_ranks).sum())}/{B} ranks identical to the reference")

    # ---- bf16 training path
    ml, tr = forward()
    e_l, rms = float((ml - ref_logits).abs().max()), float((ml - ref_logits).pow(2).mean().sqrt())
    r_t = _rel(tr.numpy(), ref_trans.numpy())
    print(f"   bf16 path  max|dlogit| {e_l:.3e}  rms {rms:.3e}  (logit scale {float(ref_logits.abs().max()):.2f})  trans rows rel-L2 {r_t:.3e}")
    st = model.store
    st.zero_grad()
    loss = lit.training_step(dict(gb), 1)
    loss.backward()
    torch.cuda.synchronize()
    dl = abs(float(loss) - float(g["loss"]))
    print(f"   loss hip {float(loss):.6f} reference {float(g['loss']):.6f}")
    ev = lit._eval(dict(gb), 0)
    amb = ((ref_logits - lab[:, None]).abs() < 2 * e_l).sum(1).numpy() - 1
    assert np.all(np.abs(ev["entity_ranks"] - ref_ranks) <= amb), (ev["entity_ranks"], ref_ranks, amb)
    if cond:
        # north_star's number, absolute: the default configuration (this is what bench.py times) within 1e-2 of the reference CPU path on
        # every one of the 66 016 logits.  For scale: the reference itself, fp32 math, with nothing but its weight matrices rounded to bf16
        # (the golden's control) moves them by max 1.02e-2 / rms 2.2e-3 -- the text stream's fp16 operands and the split-precision head put
        # this path BELOW that control (tools/error_budget.py: text stream 58 % + head 36 % of the plain-bf16 error variance).
        c_l = float(np.abs(g["ctl::mask_logits"] - ref_logits.numpy()).max())
        c_rms = float(np.sqrt(((g["ctl::mask_logits"] - ref_logits.numpy()) ** 2).mean()))
        print(f"   control (reference, bf16-rounded weight matrices): max|dlogit| {c_l:.3e} rms {c_rms:.3e}  ->  bf16 path = {e_l / c_l:.2f} x / {rms / c_rms:.2f} x")
        assert e_l < 1e-2, f"north_star: bf16 logits within 1e-2 of the reference CPU path (got {e_l:.3e})"
        assert rms < c_rms, (rms, c_rms)
        assert r_t < 1.5e-2 and dl < 5e-3
        _grad_report(st, g, tol_rel=0.12, tol_cos=0.99, tol_norm=0.08)
    else:
        # plain weights: the map is chaotic in layers 8-11 (unscaled fusion softmax); the per-layer test below shows that every
        # layer is at rounding level.  End to end, every quantity is held to 3 x its displacement in the golden's sensitivity
        # control (the reference itself with bf16-rounded weight matrices) + a small absolute term.
        c_t = _rel(g["ctl::trans_rows"], ref_trans.numpy())
        c_l = float(np.abs(g["ctl::mask_logits"] - ref_logits.numpy()).max())
        c_loss = abs(float(g["ctl::loss"]) - float(g["loss"]))
        print(f"   control: trans rows rel-L2 {c_t:.3e}, max|dlogit| {c_l:.3e}, loss moves by {c_loss:.3e}")
        # multiples: the control rounds the weight matrices ONCE; the bf16 pipeline also rounds every GEMM operand (~50 roundings
        # per layer), so its displacement is a few times the control's: 4x for L2-type quantities, 5x for maxima, 8x for the
        # scalar loss (measured: 3.0x / 3.4x / 0.05x on this batch)
        assert r_t < 4 * c_t + 1e-2 and e_l < 5 * c_l + 1e-2 and dl < 8 * c_loss + 1e-2, (r_t, e_l, dl)
        _grad_report(st, g, tol_rel=0.03, tol_cos=0.0, tol_norm=0.03, ctl_mult=4.0)


@pytest.mark.parametrize("tag", ["g7_bench_plain", "g7_bench_cond", "g8_pretrain_plain"])
def test_fp32_training_step_vs_reference(tag):
    """The fp32-accurate training step (set_precision("fp32") with gradients enabled: engine_precise.PreciseUnimoTrain) against the
    reference's own fp32 step (PL precision 32, lit_models/transformer.py:59-113) -- loss and ALL gradient tensors, including the PLAIN
    N(0,0.02) weights where the bf16 path can only be compared with a control."""
    g = _load(tag)
    pre = bool(int(g["pretrain"]))
    model, lit, cfg = _product(g, pretrain=pre)
    batch = _batch(g, pretrain=pre)
    gb = {k: v.cuda() for k, v in batch.items()}
    model.eval()
    model.set_precision("fp32")
    st = model.store
    st.zero_grad()
    loss = lit.training_step(dict(gb), 1)
    loss.backward()
    torch.cuda.synchronize()
    model.set_precision("bf16")
    dl = abs(float(loss) - float(g["loss"]))
    norms = dict(zip(g["grad_norm_names"].tolist(), g["grad_norm_vals"].tolist()))
    worst, worst_n, n_cmp = 0.0, "", 0
    aw_ref, aw_got = [], []
    for n, ref in norms.items():
        if n.endswith("decoder.weight"):
            continue
        got = float(st.g(n).double().norm())
        if ref < 1e-7:
            assert got < 1e-5, (n, got, ref)
            continue
        if "adaptive_weight" in n:          # scalars: sums of signed score x d(score) terms with heavy cancellation -> compared as ONE vector
            aw_ref.append(ref); aw_got.append(got)
            continue
        r = abs(got - ref) / ref
        n_cmp += 1
        if r > worst:
            worst, worst_n = r, n
    r_aw = _rel(aw_got, aw_ref) if aw_ref else 0.0
    rels = []
    for k in g:
        if k.startswith("gs::") and np.linalg.norm(g[k]) >= 1e-7 and "adaptive_weight" not in k:
            rels.append((_rel(_sample(st.g(k[4:])), g[k]), k[4:]))
    rels.sort(reverse=True)
    print(f"\n{tag} fp32 training step: loss hip {float(loss):.7f} reference {float(g['loss']):.7f} (|d| {dl:.2e}); worst gradient-norm deviation "
          f"{worst:.3e} ({worst_n}) over {n_cmp} tensors; adaptive-weight gradients as a vector ({len(aw_ref)}) rel-L2 {r_aw:.3e}; "
          f"worst sample rel-L2 {rels[0][0]:.3e} ({rels[0][1]}), median {rels[len(rels) // 2][0]:.3e}")
    for n in g["none_grad"].tolist():
        if n in st.slots and not n.endswith("decoder.weight"):
            assert float(st.g(n).abs().max()) == 0.0, n
    assert dl < 1e-4
    if tag == "g7_bench_plain":
        # the fine-tune network at plain weights is chaotic in the unscaled fusion softmax: the CPU oracle (the same math in another
        # summation order) is itself only within 2e-3 of these gradient norms (tests/test_oracle_vs_golden.py); the bf16 path's median
        # sample displacement on this golden is 0.55 and the reference's own bf16-weight control 1.9
        assert worst < 5e-3 and rels[0][0] < 5e-3 and r_aw < 1e-2, (worst, worst_n, rels[:3], r_aw)
    else:
        assert worst < 1e-3 and rels[0][0] < 1e-3 and r_aw < 1e-3, (worst, worst_n, rels[:3], r_aw)


def test_text_fp16_forward_vs_plain_bf16_text_stream():
    """engine.text_f16 (default): the text stream's forward products on fp16 operands.  Against MART_TEXT_F16=0 (plain bf16 text stream) on the
    conditioned G7 golden: closer to the reference, below north_star's 1e-2; the evaluation pass (no_grad) and a forward that saves
    activations for a backward pass compute the same logits bit for bit (one configuration for training and evaluation)."""
    g = _load("g7_bench_cond")
    model, lit, cfg = _product(g)
    eng = model.engine
    assert eng.text_f16 is True, "default: fp16 forward operands in the text stream"
    batch = _batch(g)
    B = int(g["B"])
    gb = {k: v.cuda() for k, v in batch.items()}
    ids = torch.tensor(cfg["analogy_entity_ids"], device="cuda")
    ar = torch.arange(B, device="cuda")
    rows = torch.from_numpy(g["trans_row_index"]).cuda()
    ref = torch.from_numpy(g["mask_logits"])
    model.eval()
    keys = ("input_ids", "attention_mask", "token_type_ids", "pixel_values", "sep_idx")

    def logits(grad):
        with torch.enable_grad() if grad else torch.no_grad():
            out, _ = model(**{k: gb[k] for k in keys}, return_dict=True)
            return out.logits[ar, rows[:, 0]][:, ids].detach().float().cpu()

    ev, tr = logits(False), logits(True)
    eng.text_f16 = False
    plain = logits(False)
    eng.text_f16 = True
    if eng.ln_fold:       # MART_LN_FOLD=1 (opt-in): no_grad passes fold the vision LayerNorms into Q/K/V and fc1 -- rounding-level differences from the training forward
        assert float((ev - tr).abs().max()) < 6e-3
    else:
        assert torch.equal(ev, tr), "evaluation and training forward passes are one configuration"
    stat = lambda x: (float((x - ref).abs().max()), float((x - ref).pow(2).mean().sqrt()))
    (e_h, r_h), (e_p, r_p) = stat(ev), stat(plain)
    print(f"\ntext stream fp16 operands: max|dlogit| {e_h:.3e} rms {r_h:.3e};  plain bf16 text stream: {e_p:.3e} / {r_p:.3e}")
    assert e_h < 1e-2 and r_h < 0.75 * r_p, (e_h, r_h, r_p)


def test_text_fp16_training_step_with_dropout_matches_the_plain_step():
    """Train mode (dropout on): the fp16 text stream draws the SAME dropout masks as the plain bf16 text stream (same seeds, same element
    indices), so with equal seeds the two steps agree to bf16 rounding: loss within 2e-2, gradients of the sampled tensors within 10 %,
    everything finite."""
    g = _load("g7_bench_cond")
    model, lit, cfg = _product(g)
    batch = _batch(g)
    gb = {k: v[:8].cuda() for k, v in batch.items()}
    st = model.store
    model.train()
    res = []
    for f16 in (False, True):
        model.engine.text_f16 = f16
        model._step = 100                                  # same dropout stream for both passes
        st.zero_grad()
        loss = lit.training_step(dict(gb), 1)
        loss.backward()
        torch.cuda.synchronize()
        assert bool(torch.isfinite(st.grad).all()) and bool(torch.isfinite(loss.detach()))
        res.append((float(loss.detach()), st.grad.clone()))
    (l0, g0), (l1, g1) = res
    names = ["unimo.encoder.text_layer.0.attention.self.query.weight", "unimo.encoder.text_layer.11.output.dense.weight",
             "unimo.encoder.vision_layers.5.mlp.fc1.weight", "cls.predictions.transform.dense.weight", "unimo.text_embeddings.word_embeddings.weight"]
    worst = 0.0
    for n in names:
        sl = st.slots[n]
        a, b = g0[sl.offset:sl.offset + sl.numel], g1[sl.offset:sl.offset + sl.numel]
        worst = max(worst, float((a - b).norm() / a.norm()))
    print(f"\ntrain-mode step, dropout on: loss plain bf16 text {l0:.5f} fp16 text {l1:.5f}; worst gradient rel-L2 difference {worst:.3e}")
    assert abs(l0 - l1) < 2e-2 and worst < 0.10


@pytest.mark.parametrize("tag,pre", [("g7_bench_cond", False), ("g7_bench_plain", False), ("g8_pretrain_cond", True)])
def test_last_layer_row_subset_equals_the_dense_path(tag, pre):
    """The trainer surface tells the model which rows of trans_hidden_states it reads (5 per example: [MASK] + the relaxation-loss rows; 1 for the
    pre-train step) and the last text layer + head transform run on those rows only (engine.forward rows=...).  Exact by construction: loss
    and ALL gradients equal the dense pass's (MART_LAST_ROWS=0) to rounding, the returned rows are the dense pass's rows, the others zero."""
    g = _load(tag)
    model, lit, cfg = _product(g, pretrain=pre)
    batch = _batch(g, pretrain=pre)
    gb = {k: v.cuda() for k, v in batch.items()}
    B, L = gb["input_ids"].shape
    st = model.store
    model.eval()
    assert lit.last_layer_rows is True
    res = []
    for on in (False, True):
        lit.last_layer_rows = on
        st.zero_grad()
        loss = lit.training_step(dict(gb), 1)
        loss.backward()
        torch.cuda.synchronize()
        res.append((float(loss.detach()), st.grad.clone()))
    (l0, g0), (l1, g1) = res
    worst, wn = 0.0, ""
    for n, sl in st.slots.items():
        a, b = g0[sl.offset:sl.offset + sl.numel], g1[sl.offset:sl.offset + sl.numel]
        na = float(a.norm())
        if na < 1e-9:
            assert float(b.abs().max()) == 0.0, n
            continue
        r = float((a - b).norm()) / na
        if r > worst:
            worst, wn = r, n
    print(f"\n{tag}: loss dense {l0:.7f} row subset {l1:.7f}; worst gradient rel-L2 difference over {len(st.slots)} tensors {worst:.3e} ({wn})")
    assert abs(l0 - l1) < 2e-6 * max(1.0, abs(l0)) and worst < 2e-3, (l0, l1, worst, wn)
    # forward values: requested rows identical to the dense pass, the others NaN
    keys = ("input_ids", "attention_mask", "token_type_ids", "pixel_values") + (() if pre else ("sep_idx",))
    need = torch.stack([gb["rel_idx"][:, 0], gb["rel_idx"][:, 1], gb["q_head_idx"]], 1) if not pre else (gb["input_ids"] == 103).int().argmax(1)[:, None]
    need = torch.cat([need, need[:, :1]], 1)                     # a repeated position: computed twice, scattered once
    need_before = need.clone()
    with torch.no_grad():
        _, dense = model(**{k: gb[k] for k in keys}, return_dict=True)
        out_p, part = model(**{k: gb[k] for k in keys}, return_dict=True, needed_rows=need)
    assert torch.equal(need, need_before)                        # the caller's tensor is not clamped in place
    ar = torch.arange(B, device="cuda")[:, None]
    d = float((dense[ar, need] - part[ar, need]).abs().max())
    mask = torch.ones(B, L, dtype=torch.bool, device="cuda")
    mask[ar, need] = False
    print(f"   requested rows: max|dense - subset| {d:.3e}; other rows all NaN: {bool(torch.isnan(part[mask]).all())}")
    assert d < 1e-5 and bool(torch.isnan(part[mask]).all())
    # the guard: a requested position scores, any other position (or the full tensor) raises instead of returning bias-only numbers
    lg = out_p.logits
    ok = lg[ar[:, 0], need[:, 0]][:, BASE:BASE + 8]
    assert bool(torch.isfinite(ok.float()).all())
    other = (need[:, 0] + 1) % L
    other = torch.where((other[:, None] == need).any(1), (other + 1) % L, other)
    other = torch.where((other[:, None] == need).any(1), (other + 1) % L, other)
    other = torch.where((other[:, None] == need).any(1), (other + 1) % L, other)
    with pytest.raises(ValueError, match="needed_rows"):
        lg[ar[:, 0], other]
    with pytest.raises(ValueError, match="needed_rows"):
        lg.materialize()
    # negative positions wrap like the reference's fancy indexing
    with torch.no_grad():
        _, neg = model(**{k: gb[k] for k in keys}, return_dict=True, needed_rows=need - L)
    assert float((neg[ar, need] - part[ar, need]).abs().max()) == 0.0


@pytest.mark.parametrize("tag", ["g8_pretrain_cond", "g8_pretrain_plain", "g8b_pretrain_p196_cond"])
def test_pretrain_step_vs_reference(tag):
    """BASELINE configs[4]: L=96, no sep_idx (no reweight; adaptive weights get no gradient), mixed pre_type, LSCE over the
    full entity / relation slices (lit_models/transformer.py:72-90,129-156)."""
    g = _load(tag)
    cond = bool(int(g["conditioned"]))
    model, lit, cfg = _product(g, pretrain=True)
    batch = _batch(g, pretrain=True)
    B = int(g["B"])
    gb = {k: v.cuda() for k, v in batch.items()}
    model.eval()
    st = model.store
    st.zero_grad()
    loss = lit.training_step(dict(gb), 1)
    loss.backward()
    torch.cuda.synchronize()
    print(f"\n{tag}: loss hip {float(loss):.6f} reference {float(g['loss']):.6f}")
    # logits of both heads on the mask rows
    with torch.no_grad():
        out, trans = model(**{k: gb[k] for k in ("input_ids", "attention_mask", "token_type_ids", "pixel_values")}, return_dict=True)
        rows = out.logits.mask_rows(gb["input_ids"], 103)
        ent = rows[:, BASE:BASE + NE].float().cpu().numpy()
        rel = rows[:, BASE + NE:BASE + NE + NR].float().cpu().numpy()
    e_e, e_r = np.abs(ent - g["entity_logits"]).max(), np.abs(rel - g["relation_logits"]).max()
    rms = float(np.sqrt(((ent - g["entity_logits"]) ** 2).mean()))
    print(f"   entity logits max|err| {e_e:.3e} rms {rms:.3e}; relation logits max|err| {e_r:.3e}")
    ev = lit._eval(dict(gb), 0)
    pt = batch["pre_type"].numpy()
    for key, ref_lg, sel in (("entity_ranks", g["entity_logits"], pt != 2), ("relation_ranks", g["relation_logits"], pt == 2)):
        ref_r = g["ranks::" + key]
        lab = batch["label"].numpy()[sel]
        lg = ref_lg[sel]
        amb = (np.abs(lg - lg[np.arange(len(lab)), lab][:, None]) < 2 * max(e_e, e_r)).sum(1) - 1
        print(f"   {key} hip {ev[key].tolist()} reference {ref_r.tolist()} ambiguous {amb.tolist()}")
        assert np.all(np.abs(ev[key] - ref_r) <= amb)
    # the tied decoder: gradient rows of entity / relation tokens that only the scoring head touches, and the decoder bias
    gw = st.g("unimo.text_embeddings.word_embeddings.weight")
    r_e = _rel(gw[BASE + 17:BASE + NE:997].float().cpu().numpy(), g["wordemb_entity_rows"])
    r_r = _rel(gw[BASE + NE:BASE + NE + NR:13].float().cpu().numpy(), g["wordemb_relation_rows"])
    r_b = _rel(st.g("cls.predictions.bias")[BASE:BASE + NE + NR].float().cpu().numpy(), g["decoder_bias_grad"])
    print(f"   tied-embedding gradient rows: entity slice rel-L2 {r_e:.3e}, relation slice {r_r:.3e}; decoder bias gradient {r_b:.3e}")
    if cond:
        assert abs(float(loss) - float(g["loss"])) < 1e-2
        c_e = float(np.abs(g["ctl::entity_logits"] - g["entity_logits"]).max())
        c_rms = float(np.sqrt(((g["ctl::entity_logits"] - g["entity_logits"]) ** 2).mean()))
        c_r = float(np.abs(g["ctl::relation_logits"] - g["relation_logits"]).max())
        print(f"   control: entity logits max {c_e:.3e} rms {c_rms:.3e}; relation logits max {c_r:.3e}")
        # north_star's 1e-2, absolute, on both pre-train heads (default configuration); and below the reference's own bf16-weight control in rms
        assert e_e < 1e-2 and e_r < 1e-2, f"north_star: bf16 logits within 1e-2 of the reference (entity {e_e:.3e}, relation {e_r:.3e})"
        assert rms < c_rms, (rms, c_rms)
        assert r_e < 0.05 and r_r < 0.05 and r_b < 0.02
        _grad_report(st, g, tol_rel=0.12, tol_cos=0.99, tol_norm=0.08)
    else:
        c_e = float(np.abs(g["ctl::entity_logits"] - g["entity_logits"]).max())
        c_loss = abs(float(g["ctl::loss"]) - float(g["loss"]))
        c_we = _rel(g["ctl::wordemb_entity_rows"], g["wordemb_entity_rows"])
        c_wr = _rel(g["ctl::wordemb_relation_rows"], g["wordemb_relation_rows"])
        c_b = _rel(g["ctl::decoder_bias_grad"], g["decoder_bias_grad"])
        print(f"   control: max|dlogit| {c_e:.3e}, loss moves by {c_loss:.3e}, tied rows {c_we:.3e} / {c_wr:.3e}, bias {c_b:.3e}")
        assert abs(float(loss) - float(g["loss"])) < 8 * c_loss + 1e-2 and e_e < 5 * c_e + 1e-2      # same multiples as the G7 test
        assert r_e < 4 * c_we + 0.02 and r_r < 4 * c_wr + 0.02 and r_b < 4 * c_b + 0.02
        _grad_report(st, g, tol_rel=0.03, tol_cos=0.0, tol_norm=0.03, ctl_mult=4.0)
    for n in st.slots:
        if "adaptive_weight" in n:
            assert float(st.g(n).abs().max()) == 0.0, n


def test_pretrain_full_size_properties():
    """configs[4] at its full size on one GPU (B=256, L=96, P=196, E=11292 / R=192 heads): the mean-loss gradient of a batch
    of two identical halves equals the gradient of one half (every split reduction / atomic / stream join of the backward pass
    incl. the 11292-wide scoring head and the sep_idx=None attention), everything finite, device ranks == host double sort."""
    import bench
    from mkg_analogy_amd import data_synth as D
    dev = torch.device("cuda", 0)
    model, lit, cfg = bench.build(16, seed=0, device=dev, backbone="mkgformer")
    lit.args.pretrain = 1
    model.eval()
    half = D.make_batch(128, 96, seed=555, device=dev, pretrain=True)
    full = {k: torch.cat([v, v], 0) for k, v in half.items()}
    grads = []
    for bt in (half, full):
        model.store.zero_grad()
        loss = lit.training_step(dict(bt), 0)
        loss.backward()
        torch.cuda.synchronize()
        assert bool(torch.isfinite(model.store.grad).all())
        grads.append((float(loss.detach()), model.store.grad.clone()))
    (l1, g1), (l2, g2) = grads
    rel = float((g1 - g2).norm() / g1.norm())
    print(f"\npretrain loss half {l1:.6f} full {l2:.6f}; gradient rel-L2 difference {rel:.2e}")
    assert abs(l1 - l2) < 1e-5 * max(1.0, abs(l1)) and rel < 1e-3
    ev = lit._eval(dict(full), 0)
    with torch.no_grad():
        out, _ = model(**{k: full[k] for k in ("input_ids", "attention_mask", "token_type_ids", "pixel_values")}, return_dict=True)
        rows = out.logits.mask_rows(full["input_ids"], 103)
        pt = full["pre_type"]
        for key, sel, (a, b) in (("entity_ranks", pt != 2, (BASE, BASE + NE)), ("relation_ranks", pt == 2, (BASE + NE, BASE + NE + NR))):
            idx = sel.nonzero(as_tuple=True)[0]
            lg = rows[idx, a:b].float().cpu()
            lab = full["label"][idx].cpu()
            order = torch.argsort(lg, dim=1, descending=True, stable=True)
            host = (torch.argsort(order, dim=1, stable=True)[torch.arange(len(lab)), lab] + 1).numpy()
            untied = ((lg == lg[torch.arange(len(lab)), lab][:, None]).sum(1) == 1).numpy()
            assert untied.sum() > 0.9 * len(lab)
            assert np.array_equal(np.asarray(ev[key])[untied], host[untied]), key


def test_teacher_forced_layers_plain_weights():
    """Per-layer parity on PLAIN N(0,0.02) weights at real dimensions (P=196, B=2).  Every layer of the HIP engine is fed the
    ORACLE's inputs of that layer (engine.inject) and the oracle's upstream gradients (engine.inject_grad), so the error measured
    at a layer is its own bf16 rounding, not what the chaotic fusion softmax of layers 8-11 makes of earlier roundings.  The
    bound for a layer's output is 3 x the displacement of the fp32 oracle layer when only its inputs and weight matrices are
    rounded to bf16 (the intrinsic sensitivity of that layer's math) + 5e-3; parameter and input gradients are asserted per layer."""
    from mkg_analogy_amd import data_synth as D
    g = dict(patch=np.int64(16), weight_seed=np.int64(0), conditioned=np.int64(0))
    model, lit, cfg = _product(g)
    vc, tc = O.VisionCfg(patch_size=16), O.TextCfg(vocab_size=BASE + NE + NR + 1)
    sd = O.init_relation_word(O.init_params(vc, O.TextCfg(vocab_size=BASE + NE + NR), seed=0), cfg["analogy_relation_ids"])
    B, L = 2, 64
    batch = D.make_batch(B, L, seed=77)
    ids = torch.tensor(cfg["analogy_entity_ids"])
    # ---- oracle with taps and their gradients
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    taps = {}
    vis0 = O.vision_embed(sdg, vc, batch["pixel_values"])
    txt0 = O.text_embed(sdg, tc, batch["input_ids"], batch["token_type_ids"], False)
    ext = O.extended_mask(batch["attention_mask"])
    seq = O.encoder(sdg, vc, tc, vis0, txt0, ext, batch["sep_idx"], False, taps)
    trans_ref = O.head_transform(sdg, tc, seq)
    taps["vis_emb"], taps["txt_emb"] = vis0, txt0
    for t in taps.values():
        t.retain_grad()
    loss_ref, _ = O.finetune_loss(sdg, trans_ref, batch["input_ids"], batch["label"], batch["rel_idx"], batch["q_head_idx"], batch["a_head_idx"], ids, alpha=0.43)
    loss_ref.backward()
    # ---- HIP engine, teacher-forced at every layer boundary
    eng = model.engine
    eng.two_stream = eng.overlap_wgrad = False
    eng.taps = {}
    n = vc.num_hidden_layers
    eng.inject = {k: v.detach() for k, v in taps.items()}
    eng.inject_grad = {k: v.grad for k, v in taps.items() if k not in ("vis_emb", "txt_emb") and v.grad is not None and k != f"txt{n - 1}"}
    model.eval()
    gb = {k: v.cuda() for k, v in batch.items()}
    st = model.store
    st.zero_grad()
    loss = lit.training_step(dict(gb), 1)
    loss.backward()
    torch.cuda.synchronize()
    got = {k: v.detach().float().cpu() for k, v in eng.taps.items()}
    eng.taps = eng.inject = eng.inject_grad = None

    # intrinsic bf16 sensitivity of each oracle layer (inputs + weight matrices rounded once, fp32 math)
    rb = lambda x: x.detach().to(torch.bfloat16).float()
    sdb = {k: (rb(v) if v.dim() >= 2 and "embeddings" not in k else v.detach()) for k, v in sd.items()}
    print()
    with torch.no_grad():
        kv = None
        for l in range(n):
            vin = taps["vis_emb"] if l == 0 else taps[f"vis{l - 1}"]
            tin = taps["txt_emb"] if l == 0 else taps[f"txt{l - 1}"]
            v_ctl = O.vision_layer(sdb, vc, l, rb(vin), kv if l >= 8 else None)
            t_ctl, kv_new = O.text_layer(sdb, tc, l, rb(tin), ext, batch["sep_idx"], rb(taps[f"vis{l}"]) if l >= 8 else None, False)
            kv = kv_new if l >= 7 else None
            for name, ctl in ((f"vis{l}", v_ctl), (f"txt{l}", t_ctl)):
                ref = taps[name].detach()
                s_ctl = float((ctl - ref).norm() / ref.norm())
                err = float((got[name] - ref).norm() / ref.norm())
                print(f"   layer out {name:6s}: HIP rel-L2 {err:.3e}   one-rounding control {s_ctl:.3e}")
                assert err < 3.0 * s_ctl + 5e-3, (name, err, s_ctl)
    # gradients w.r.t. the streams one layer up, and the parameter gradients of every layer
    for l in range(n - 1):
        for name in (f"vis{l}", f"txt{l}"):
            ref = taps[name].grad
            err = float((got["d" + name] - ref).norm() / (ref.norm() + 1e-30))
            print(f"   d(loss)/d({name}) from teacher-forced layer {l + 1}: rel-L2 {err:.3e}")
            assert err < (0.03 if l < 7 else 0.10), (name, err)
    aw = sorted(k for k in sdg if "adaptive_weight" in k)
    ref = np.array([float(sdg[k].grad) for k in aw])
    hip = np.array([float(st.g(k)) for k in aw])
    print(f"   adaptive-weight gradients, all {len(aw)} scalars as one vector: rel-L2 {_rel(hip, ref):.3e}  (|ref| max {np.abs(ref).max():.2e}, min {np.abs(ref).min():.2e})")
    assert _rel(hip, ref) < 0.15
    worst = 0.0
    for l in range(n):
        for name, p in sdg.items():
            if (f"vision_layers.{l}." not in name and f"text_layer.{l}." not in name) or "adaptive_weight" in name:
                continue
            if p.grad is None or float(p.grad.norm()) < 1e-7:
                continue
            gh = st.g(name).detach().float().cpu()
            err = float((gh - p.grad).norm() / p.grad.norm())
            worst = max(worst, err)
            tol = 0.05 if l < 7 else 0.15
            if err > 0.5 * tol:
                print(f"   param grad {name}: rel-L2 {err:.3e}")
            assert err < tol, (name, err)
    print(f"   worst per-layer parameter-gradient rel-L2 (teacher-forced): {worst:.3e}")


def test_full_bench_batch_forward_vs_oracle():
    """BASELINE configs[1] at its FULL size -- all 256 examples of bench.py's rank-0 batch, P=196, L=64 (100 608 vision tokens: every
    tile of every launch is a full-size tile of the benchmark) -- forward pass of the bf16 training path and of the fp32-accurate
    path against the fp32 CPU oracle run on this box's host cores (no gradient: ~1 minute).  The oracle is pinned to the reference on
    the first 32 examples of this very batch by G7 (tests/test_oracle_vs_golden.py); conditioned weights, as in G7's absolute test."""
    g = _load("g7_bench_cond")
    model, lit, cfg = _product(g)
    from mkg_analogy_amd import data_synth as D
    B, L = int(g["batch_total"]), int(g["L"])
    assert B == 256
    batch = D.make_batch(B, L, seed=int(g["batch_seed"]))
    ids = torch.tensor(cfg["analogy_entity_ids"])
    vc = O.VisionCfg(patch_size=int(g["patch"]))
    tc = O.TextCfg(vocab_size=BASE + NE + NR + 1)
    sd = O.init_relation_word(O.condition_weights(O.init_params(vc, O.TextCfg(vocab_size=BASE + NE + NR), seed=int(g["weight_seed"]))),
                              cfg["analogy_relation_ids"])
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    with torch.no_grad():
        ml_ref = []
        for s in range(0, B, 32):                                   # the oracle in slices (its activations are fp32 on the host); per-example independent in eval mode
            sl = slice(s, s + 32)
            _, tr = O.forward(sd, vc, tc, batch["input_ids"][sl], batch["attention_mask"][sl], batch["token_type_ids"][sl],
                              batch["pixel_values"][sl], batch["sep_idx"][sl], train=False)
            _, ml = O.finetune_loss(sd, tr, batch["input_ids"][sl], batch["label"][sl], batch["rel_idx"][sl], batch["q_head_idx"][sl],
                                    batch["a_head_idx"][sl], ids, alpha=0.43)
            ml_ref.append(ml.float())
        ml_ref = torch.cat(ml_ref)
    assert float((ml_ref[:32] - torch.from_numpy(g["mask_logits"])).abs().max()) < 2e-4          # the slice G7 pins to the reference
    gb = {k: v.cuda() for k, v in batch.items()}
    ar = torch.arange(B, device="cuda")
    _, mi = (gb["input_ids"] == 103).nonzero(as_tuple=True)
    model.eval()
    keys = ("input_ids", "attention_mask", "token_type_ids", "pixel_values", "sep_idx")

    def forward():
        with torch.no_grad():
            out, _ = model(**{k: gb[k] for k in keys}, return_dict=True)
            return out.logits[ar, mi][:, ids.cuda()].float().cpu()

    ml = forward()
    e_l, rms = float((ml - ml_ref).abs().max()), float((ml - ml_ref).pow(2).mean().sqrt())
    model.engine.text_f16 = False                               # plain bf16 text stream (MART_TEXT_F16=0), for scale
    mls = forward()
    model.engine.text_f16 = True
    e_s, rms_s = float((mls - ml_ref).abs().max()), float((mls - ml_ref).pow(2).mean().sqrt())
    model.set_precision("fp32")
    ml32 = forward()
    model.set_precision("bf16")
    e32 = float((ml32 - ml_ref).abs().max())
    rank = lambda x: (x > x[torch.arange(B), batch["label"]][:, None]).sum(1) + 1
    same32 = int((rank(ml32) == rank(ml_ref)).sum())
    # the reference's own one-rounding control exists for the first 32 examples (G7): the max norm is compared on that slice, the rms
    # (independent of the sample size) on all 528 128 logits
    ctl = torch.from_numpy(g["ctl::mask_logits"]) - torch.from_numpy(g["mask_logits"])
    c_l, c_rms = float(ctl.abs().max()), float(ctl.pow(2).mean().sqrt())
    e_l32 = float((ml[:32] - ml_ref[:32]).abs().max())
    print(f"\nB=256 P=196: bf16 path max|dlogit| {e_l:.3e} (first 32 examples {e_l32:.3e}) rms {rms:.3e}  [control on the first 32: {c_l:.3e} / {c_rms:.3e}]; "
          f"plain bf16 text stream max {e_s:.3e} rms {rms_s:.3e}; fp32-accurate path max|dlogit| {e32:.3e}, {same32}/256 ranks identical "
          f"(logit scale {float(ml_ref.abs().max()):.2f})")
    assert e32 < 1e-3
    assert e_l < 1e-2, f"north_star's 1e-2 on bf16 logits, default configuration, all 256 examples x 2063 entities (got {e_l:.3e})"
    assert rms < c_rms and rms < rms_s, (rms, c_rms, rms_s)
    lab = ml_ref[torch.arange(B), batch["label"]]
    near = ((ml_ref - lab[:, None]).abs() < 2 * e32).sum(1) - 1
    assert bool(((rank(ml32) - rank(ml_ref)).abs() <= near).all())


def test_deferred_layernorm_reduction_schedule_is_bit_identical():
    """engine.ln_defer (MART_LN_DEFER=1): the dgamma / dbeta reductions run on the weight-gradient stream; the sums are the same sums."""
    g = _load("g7_bench_cond")
    model, lit, cfg = _product(g)
    batch = _batch(g)
    gb = {k: v[:16].cuda() for k, v in batch.items()}
    st = model.store
    model.eval()
    res = []
    for defer in (False, True):
        model.engine.ln_defer = defer
        st.zero_grad()
        loss = lit.training_step(dict(gb), 1)
        loss.backward()
        torch.cuda.synchronize()
        res.append((float(loss.detach()), st.grad.clone()))
    model.engine.ln_defer = False
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])


def test_fusion_side_buffer_schedule_equals_the_in_place_one():
    """engine.fusion_side (MART_FUSION_SIDE=1): the fusion op's d(visual) of text layers 8-10 goes through a side buffer that the LayerNorm-1
    backward of the vision layer above adds as a second residual operand, instead of being accumulated into the vision-stream gradient in place.
    Same gradients up to the order of two f32 additions."""
    g = _load("g7_bench_cond")
    model, lit, cfg = _product(g)
    batch = _batch(g)
    gb = {k: v[:16].cuda() for k, v in batch.items()}
    st = model.store
    model.eval()
    res = []
    for side in (False, True):
        model.engine.fusion_side = side
        st.zero_grad()
        loss = lit.training_step(dict(gb), 1)
        loss.backward()
        torch.cuda.synchronize()
        res.append((float(loss.detach()), st.grad.clone()))
    model.engine.fusion_side = False
    (l0, g0), (l1, g1) = res
    worst, wn = 0.0, ""
    for n, sl in st.slots.items():
        a, b = g0[sl.offset:sl.offset + sl.numel], g1[sl.offset:sl.offset + sl.numel]
        na = float(a.norm())
        if na < 1e-9 or sl.numel < 768 or n.endswith(("k_proj.bias", "key.bias")):
            # key biases: zero gradient in exact arithmetic (softmax shift invariance), rounding noise here; the adaptive attention weights are
            # scalars summed over every score with heavy cancellation: one flipped bf16 rounding upstream moves them by percents (measured 11 %)
            continue
        r = float((a - b).norm()) / na
        if r > worst:
            worst, wn = r, n
    print(f"\nfusion d(visual) in place vs side buffer: loss {l0:.7f} / {l1:.7f}; worst gradient rel-L2 difference {worst:.3e} ({wn})")
    # the two schedules differ by the order of two f32 additions; a bf16 rounding that flips downstream of it travels through eleven layers of
    # backward pass: 5.7e-3 rel-L2 on a text layer 1 weight was measured, the same order as the bf16 path's own distance to the reference (1e-2)
    assert l0 == l1 and worst < 2e-2
