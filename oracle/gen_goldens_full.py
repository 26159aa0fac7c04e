"""G7 / G8: golden vectors of the UNMODIFIED reference at the benchmark's own shapes.

Run in the build container only (the reference does not exist on the GPU box):

    python oracle/gen_goldens_full.py            # writes tests/golden/g7_*.npz, g8_*.npz

TEST INFRASTRUCTURE ONLY (same rules as gen_goldens.py: the reference modules are imported in place, no reference
source is copied, only inputs + outputs are stored).

G7  BASELINE configs[1] shape: the first 32 examples of bench.py's rank-0 batch (data_synth.make_batch(256, 64,
    seed=1234)), ViT-B/16 patches (P=196, 393 vision tokens), L=64, V=42007, eval mode (dropout off), fp32.
    Two weight sets regenerated from a numpy seed on every side: "plain" (N(0,0.02), what bench.py's network looks
    like: chaotic in the unscaled fusion softmax) and "cond" (oracle.condition_weights: same network, smooth map).
    Stored: mask logits [32,2063], trans_hidden at the five rows the loss reads, loss, ranks, metrics, all 451
    gradient norms, strided samples of ~45 gradient tensors, and per-layer hidden states (reference forward hooks)
    of layers 0, 7, 8, 11 for examples 0-1 (pins the oracle's per-layer taps at real dimensions).
G8b the same pre-train step at P=196 (ViT-B/16 patches, 393 vision tokens), B=8, conditioned weights.
G9  BASELINE configs[3]: the reference's FlavaForMaskedLM at real dimensions, B=2 (see g9_flava); G9b: B=8 with per-layer taps (g9b_flava).
G8  BASELINE configs[4] shape: MarKG pre-train step, L=96, no sep_idx, mixed pre_type, B=8, CLIP-B/32 patches
    (P=49, the geometry of the reference's pre-train script), full E=11292 / R=192 heads: loss, entity and relation
    ranks, gradient norms and samples (incl. the tied word-embedding rows of the entity slice and cls.predictions.bias).
"""
from __future__ import annotations

import argparse as ap
import os
import sys
import time

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import mkgformer_oracle as O  # noqa: E402
from oracle.gen_goldens import FakeTokenizer, hf_configs, load_into, load_reference  # noqa: E402
from mkg_analogy_amd import data_synth as D  # noqa: E402  (synthetic batch generator shared with bench.py)

BASE, NE, NR = D.BASE_VOCAB, D.N_ENT, D.N_REL
TAP_LAYERS = (0, 7, 8, 11)


def sample_names():
    names = ["cls.predictions.transform.dense.weight", "cls.predictions.transform.LayerNorm.weight", "cls.predictions.bias",
             "unimo.text_embeddings.word_embeddings.weight", "unimo.text_embeddings.position_embeddings.weight",
             "unimo.text_embeddings.token_type_embeddings.weight", "unimo.text_embeddings.LayerNorm.weight",
             "unimo.vision_embeddings.patch_embedding.weight", "unimo.vision_embeddings.class_embedding",
             "unimo.vision_embeddings.position_embedding.weight", "unimo.vision_pre_layrnorm.bias"]
    for l in (0, 7, 8, 11):
        t, v = f"unimo.encoder.text_layer.{l}.", f"unimo.encoder.vision_layers.{l}."
        names += [t + "attention.self.query.weight", t + "attention.self.key.weight", t + "attention.self.value.bias",
                  t + "attention.self.adaptive_weight.0", t + "attention.self.adaptive_weight.1",
                  t + "attention.output.dense.weight", t + "intermediate.dense.weight", t + "output.dense.weight",
                  t + "output.LayerNorm.weight",
                  v + "self_attn.q_proj.weight", v + "self_attn.v_proj.weight", v + "self_attn.out_proj.bias",
                  v + "layer_norm1.weight", v + "mlp.fc1.weight", v + "mlp.fc2.weight", v + "layer_norm2.bias"]
        if l >= 8:
            names += [t + "intermediate.fusion_dense.weight"]
    return names


def grad_sample(g: torch.Tensor, n: int = 1024) -> np.ndarray:
    """A strided sample of a gradient tensor (the same rule is applied to the HIP gradients in the tests)."""
    f = g.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].numpy().copy()


def build_reference(lit, unimo, patch, sd0, pretrain, cfg):
    vc, tc0 = O.VisionCfg(patch_size=patch), O.TextCfg(vocab_size=BASE)
    hv, ht = hf_configs(vc, tc0)
    torch.manual_seed(0)
    model = unimo.UnimoForMaskedLM(hv, ht)
    tok = FakeTokenizer(BASE + NE + NR)
    args = ap.Namespace(label_smoothing=0.1, alpha=0.43, pretrain=int(pretrain), lr=5e-5, weight_decay=0.01, optimizer="AdamW",
                        warm_up_radio=0.1)
    lm = lit.TransformerLitModel(model=model, args=args, tokenizer=tok, data_config=cfg)      # resize -> 42006
    load_into(model, sd0)
    lm._init_relation_word()                                                                   # -> 42007
    assert model.get_input_embeddings().weight.shape[0] == D.VOCAB and tok.extra["[R]"] == D.R_TOKEN
    model.eval()
    return model, lm


def run(lit, unimo, out_dir, tag, patch, B, L, pretrain, conditioned, weight_seed, batch_seed, batch_total, taps):
    t0 = time.time()
    cfg = D.data_config(seed=1234)
    vc = O.VisionCfg(patch_size=patch)
    sd0 = O.init_params(vc, O.TextCfg(vocab_size=BASE + NE + NR), seed=weight_seed)
    if conditioned:
        sd0 = O.condition_weights(sd0)
    full = D.make_batch(batch_total, L, seed=batch_seed, pretrain=pretrain)
    batch = {k: v[:B].clone() for k, v in full.items()}
    del full
    res = evaluate(lit, unimo, patch, sd0, pretrain, cfg, batch, B, taps)
    if True:
        # sensitivity control: the SAME reference run with only its weight matrices rounded to bf16 (fp32 math otherwise).  How far
        # that moves every output is the intrinsic bf16 sensitivity of the network at these weights; the GPU tests hold the
        # bf16 HIP path to a small multiple of it instead of to hand-picked numbers.  (Round 3: also for the conditioned weights,
        # so that the distance of the bf16 path from its floor is a number there too.)
        sdb = {k: (v.to(torch.bfloat16).float() if v.dim() >= 2 and "embeddings" not in k else v) for k, v in sd0.items()}
        ctl = evaluate(lit, unimo, patch, sdb, pretrain, cfg, batch, B, False)
        for k, v in ctl.items():
            if k.startswith(("gs::", "grad_norm_vals", "loss", "mask_logits", "trans_rows", "entity_logits", "relation_logits",
                             "wordemb_", "decoder_bias_grad")):
                res["ctl::" + k] = v
    ints = {"in::" + k: v.numpy() for k, v in batch.items() if k != "pixel_values"}
    pix = batch["pixel_values"]
    np.savez_compressed(
        os.path.join(out_dir, f"{tag}.npz"),
        patch=np.int64(patch), B=np.int64(B), L=np.int64(L), pretrain=np.int64(pretrain), conditioned=np.int64(conditioned),
        weight_seed=np.int64(weight_seed), batch_seed=np.int64(batch_seed), batch_total=np.int64(batch_total),
        pixel_sum=np.float64(float(pix.double().sum())), pixel_abs_sum=np.float64(float(pix.double().abs().sum())),
        **ints, **res)
    print(f"{tag}: loss {float(res['loss']):.6f} ranks { {k: v for k, v in res.items() if k.startswith('ranks::')} } ({time.time() - t0:.1f} s)", flush=True)


def evaluate(lit, unimo, patch, sd0, pretrain, cfg, batch, B, taps):
    model, lm = build_reference(lit, unimo, patch, sd0, pretrain, cfg)

    hooks, tapped = [], {}
    if taps:
        enc = model.unimo.encoder
        for l in TAP_LAYERS:
            hooks.append(enc.vision_layers[l].register_forward_hook(
                lambda m, i, o, l=l: tapped.__setitem__(f"vis{l}", o[0].detach()[:2, ::8].clone())))
            hooks.append(enc.text_layer[l].register_forward_hook(
                lambda m, i, o, l=l: tapped.__setitem__(f"txt{l}", o[0].detach()[:2, ::2].clone())))

    def clone_batch():
        return {k: v.clone() for k, v in batch.items()}

    model.zero_grad()
    loss = lm.training_step(clone_batch(), 1)
    loss.backward()
    for h in hooks:
        h.remove()
    named = dict(model.named_parameters())
    none_grad = sorted(n for n, p in named.items() if p.grad is None)
    norms = {n: float(p.grad.double().norm()) for n, p in named.items() if p.grad is not None}
    samples = {"gs::" + n: grad_sample(named[n].grad) for n in sample_names() if named[n].grad is not None}
    extra = {}
    if pretrain:
        # rows of the tied word embedding inside the entity slice that received decoder gradient only (never an input token)
        g = named["unimo.text_embeddings.word_embeddings.weight"].grad
        extra["wordemb_entity_rows"] = g[BASE + 17:BASE + NE:997].numpy().copy()
        extra["wordemb_relation_rows"] = g[BASE + NE:BASE + NE + NR:13].numpy().copy()
        extra["decoder_bias_grad"] = named["cls.predictions.bias"].grad[BASE:BASE + NE + NR].numpy().copy()

    # forward outputs the trainer reads (no_grad second pass, full logits as the reference builds them)
    fb = clone_batch()
    for k in ("label", "rel_label", "rel_idx", "q_head_idx", "a_head_idx", "pre_type"):
        fb.pop(k, None)
    with torch.no_grad():
        out, trans = model(**fb, return_dict=True)
    ar = torch.arange(B)
    _, mask_idx = (batch["input_ids"] == 103).nonzero(as_tuple=True)
    mask_rows = out.logits[ar, mask_idx]
    if pretrain:
        outs = dict(entity_logits=mask_rows[:, BASE:BASE + NE].numpy(), relation_logits=mask_rows[:, BASE + NE:BASE + NE + NR].numpy(),
                    trans_mask=trans[ar, mask_idx].numpy())
    else:
        ids = torch.tensor(cfg["analogy_entity_ids"])
        rows = torch.stack([mask_idx, batch["q_head_idx"], batch["a_head_idx"], batch["rel_idx"][:, 0], batch["rel_idx"][:, 1]], 1)
        outs = dict(mask_logits=mask_rows[:, ids].numpy(), trans_rows=trans[ar[:, None], rows].numpy(), trans_row_index=rows.numpy())
    del out
    ev = lm._eval(clone_batch(), 0)
    ranks = {k: np.asarray(v) for k, v in ev.items()}
    if not pretrain:
        lm.validation_epoch_end([ev])
        mets = dict(lm.__dict__["_logged"])
        extra["metric_names"] = np.array(sorted(mets))
        extra["metric_vals"] = np.array([mets[k] for k in sorted(mets)])

    out = dict(loss=np.float64(float(loss.detach())), none_grad=np.array(none_grad),
               grad_norm_names=np.array(sorted(norms)), grad_norm_vals=np.array([norms[k] for k in sorted(norms)]))
    out.update({"ranks::" + k: v for k, v in ranks.items()})
    out.update({"tap::" + k: v.numpy() for k, v in tapped.items()})
    out.update(outs); out.update(samples); out.update(extra)
    return out


def g9_flava(out_dir):
    """G9 (round 3): the reference's FlavaForMaskedLM at REAL dimensions (flava-full: 768 wide, 12 + 12 + 6 layers, 393 image
    tokens, L = 64, vocabulary 42007), B = 2, eval mode, fine-tune loss (alpha 0.45, scripts/run_finetune_flava.sh).  Weights
    from flava_oracle.init_params(seed 13) + the [R] row = mean of the analogy-relation rows (lit_models/transformer.py:41-54),
    exactly what tests/test_flava_gpu.py builds on the HIP side.  Stored: trans_hidden rows the loss reads, mask logits over the
    2063 analogy entities, loss, every gradient norm, strided samples of ~70 gradient tensors."""
    from oracle import flava_oracle as FO
    from oracle.gen_goldens import load_reference_flava
    from transformers import FlavaConfig
    t0 = time.time()
    fl = load_reference_flava()
    cfgd = D.data_config(seed=1234)
    c0 = FO.FlavaCfg(vocab_size=D.VOCAB - 1)
    sd0 = FO.init_params(c0, seed=13)
    W = sd0["flava.text_model.embeddings.word_embeddings.weight"]
    sd = dict(sd0)
    sd["flava.text_model.embeddings.word_embeddings.weight"] = torch.cat([W, W[torch.tensor(cfgd["analogy_relation_ids"])].mean(0, keepdim=True)], 0)
    sd["cls.bias"] = torch.cat([sd0["cls.bias"], torch.zeros(1)])
    cfg = FlavaConfig(text_config=dict(vocab_size=D.VOCAB), image_config=dict(), multimodal_config=dict())
    assert cfg.text_config.hidden_size == 768 and cfg.image_config.patch_size == 16 and cfg.multimodal_config.num_hidden_layers == 6
    torch.manual_seed(0)
    model = fl.FlavaForMaskedLM(cfg)
    model.cls.decoder.weight = model.flava.text_model.embeddings.word_embeddings.weight      # tie manually (no resize under 5.x)
    model.flava.text_model.embeddings.word_embeddings.padding_idx = None                      # 4.19.0 resize drops padding_idx (see g5)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(("position_ids" in m or "decoder" in m or "token_type_ids" in m) for m in missing), missing
    named = dict(model.named_parameters())
    assert set(named) == set(sd), (set(named) ^ set(sd))
    model.eval()
    B, L = 2, 64
    batch = D.make_batch(B, L, seed=17)
    ids = torch.tensor(cfgd["analogy_entity_ids"])
    out, trans = model(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], token_type_ids=batch["token_type_ids"],
                       pixel_values=batch["pixel_values"], sep_idx=batch["sep_idx"], return_dict=True)
    ar = torch.arange(B)
    _, mask_idx = (batch["input_ids"] == 103).nonzero(as_tuple=True)
    mask_logits = out.logits[ar, mask_idx][:, ids]
    loss = O.label_smooth_ce(mask_logits, batch["label"], 0.1) + 0.45 * O.relaxation_loss(trans, batch["rel_idx"], batch["q_head_idx"],
                                                                                          batch["a_head_idx"])
    loss.backward()
    rows = torch.stack([mask_idx, batch["q_head_idx"], batch["a_head_idx"], batch["rel_idx"][:, 0], batch["rel_idx"][:, 1]], 1)
    none_grad = sorted(n for n, p in named.items() if p.grad is None)
    norms = {n: float(p.grad.double().norm()) for n, p in named.items() if p.grad is not None}
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
    samples = {"gs::" + n: grad_sample(named[n].grad) for n in names if named[n].grad is not None}
    ranks = O.ranks_double_sort(mask_logits.detach(), batch["label"])
    pix = batch["pixel_values"]
    np.savez_compressed(
        os.path.join(out_dir, "g9_flava_real.npz"), B=np.int64(B), L=np.int64(L), weight_seed=np.int64(13), batch_seed=np.int64(17),
        pixel_sum=np.float64(float(pix.double().sum())), pixel_abs_sum=np.float64(float(pix.double().abs().sum())),
        **{"in::" + k: v.numpy() for k, v in batch.items() if k != "pixel_values"},
        mask_logits=mask_logits.detach().numpy(), trans_rows=trans.detach()[ar[:, None], rows].numpy(), trans_row_index=rows.numpy(),
        loss=np.float64(float(loss.detach())), ranks=np.asarray(ranks), none_grad=np.array(none_grad),
        grad_norm_names=np.array(sorted(norms)), grad_norm_vals=np.array([norms[k] for k in sorted(norms)]), **samples)
    print(f"g9_flava_real: loss {float(loss):.6f} ranks {np.asarray(ranks).tolist()} none-grad {len(none_grad)} ({time.time() - t0:.1f} s)", flush=True)


G9B_TAPS = (("text_model", (0, 6, 11)), ("image_model", (0, 11)), ("multimodal_model", (0, 5)))


def g9b_tap_rows(S: int) -> np.ndarray:
    """Token rows of a tap that the golden keeps (the streams are [8, S, 768] f32: 2.4 / 9.6 / 11 MB each in full): the first two, the last one, every 16th."""
    return np.unique(np.concatenate([np.arange(0, S, 16), np.array([0, 1, S - 1])]))


def g9b_flava(out_dir):
    """G9b (round 4): G9 at B = 8 (first 8 examples of the B=8 batch of seed 17) with PER-LAYER taps -- the outputs of text layers 0 / 6 / 11, image
    layers 0 / 11 and multimodal layers 0 / 5 of the reference (forward hooks on its FlavaLayer modules; token rows g9b_tap_rows) -- next to what G9
    stores (trans_hidden rows, mask logits, loss, ranks, every gradient norm, strided gradient samples).  Same weights as G9 (seed 13)."""
    from oracle import flava_oracle as FO
    from oracle.gen_goldens import load_reference_flava
    from transformers import FlavaConfig
    t0 = time.time()
    fl = load_reference_flava()
    cfgd = D.data_config(seed=1234)
    c0 = FO.FlavaCfg(vocab_size=D.VOCAB - 1)
    sd0 = FO.init_params(c0, seed=13)
    W = sd0["flava.text_model.embeddings.word_embeddings.weight"]
    sd = dict(sd0)
    sd["flava.text_model.embeddings.word_embeddings.weight"] = torch.cat([W, W[torch.tensor(cfgd["analogy_relation_ids"])].mean(0, keepdim=True)], 0)
    sd["cls.bias"] = torch.cat([sd0["cls.bias"], torch.zeros(1)])
    cfg = FlavaConfig(text_config=dict(vocab_size=D.VOCAB), image_config=dict(), multimodal_config=dict())
    torch.manual_seed(0)
    model = fl.FlavaForMaskedLM(cfg)
    model.cls.decoder.weight = model.flava.text_model.embeddings.word_embeddings.weight
    model.flava.text_model.embeddings.word_embeddings.padding_idx = None
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    named = dict(model.named_parameters())
    assert set(named) == set(sd), (set(named) ^ set(sd))
    model.eval()
    B, L = 8, 64
    batch = D.make_batch(B, L, seed=17)
    taps = {}
    hooks = []
    for mod, ls in G9B_TAPS:
        for l in ls:
            layer = getattr(model.flava, mod).encoder.layer[l]
            hooks.append(layer.register_forward_hook(lambda m, i, o, key=f"{mod[0]}{l}": taps.__setitem__(key, (o[0] if isinstance(o, tuple) else o).detach())))
    ids = torch.tensor(cfgd["analogy_entity_ids"])
    out, trans = model(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], token_type_ids=batch["token_type_ids"],
                       pixel_values=batch["pixel_values"], sep_idx=batch["sep_idx"], return_dict=True)
    for h in hooks:
        h.remove()
    ar = torch.arange(B)
    _, mask_idx = (batch["input_ids"] == 103).nonzero(as_tuple=True)
    mask_logits = out.logits[ar, mask_idx][:, ids]
    loss = O.label_smooth_ce(mask_logits, batch["label"], 0.1) + 0.45 * O.relaxation_loss(trans, batch["rel_idx"], batch["q_head_idx"],
                                                                                          batch["a_head_idx"])
    loss.backward()
    rows = torch.stack([mask_idx, batch["q_head_idx"], batch["a_head_idx"], batch["rel_idx"][:, 0], batch["rel_idx"][:, 1]], 1)
    none_grad = sorted(n for n, p in named.items() if p.grad is None)
    norms = {n: float(p.grad.double().norm()) for n, p in named.items() if p.grad is not None}
    names = ["cls.transform.dense.weight", "cls.bias", "flava.text_model.embeddings.word_embeddings.weight",
             "flava.image_model.embeddings.position_embeddings", "flava.image_model.embeddings.patch_embeddings.projection.weight",
             "flava.multimodal_model.cls_token", "flava.image_to_mm_projection.weight", "flava.text_to_mm_projection.weight",
             "flava.multimodal_model.layernorm.weight", "flava.text_model.embeddings.LayerNorm.bias"]
    for mod, ls in G9B_TAPS:
        for l in ls:
            p = f"flava.{mod}.encoder.layer.{l}."
            names += [p + "attention.attention.query.weight", p + "attention.attention.value.weight", p + "attention.output.dense.weight",
                      p + "layernorm_before.weight", p + "intermediate.dense.weight", p + "output.dense.weight", p + "layernorm_after.bias"]
            if mod == "text_model":
                names += [p + "attention.attention.adaptive_weight.0", p + "attention.attention.adaptive_weight.1"]
    samples = {"gs::" + n: grad_sample(named[n].grad) for n in names if named[n].grad is not None}
    tap_out = {}
    for k, v in taps.items():
        assert v.shape[0] == B and v.shape[2] == 768, (k, v.shape)
        r = g9b_tap_rows(v.shape[1])
        tap_out["tap::" + k] = v[:, torch.from_numpy(r)].numpy()
        tap_out["tapnorm::" + k] = np.float64(float(v.double().norm()))
    ranks = O.ranks_double_sort(mask_logits.detach(), batch["label"])
    pix = batch["pixel_values"]
    np.savez_compressed(
        os.path.join(out_dir, "g9b_flava_b8.npz"), B=np.int64(B), L=np.int64(L), weight_seed=np.int64(13), batch_seed=np.int64(17),
        pixel_sum=np.float64(float(pix.double().sum())), pixel_abs_sum=np.float64(float(pix.double().abs().sum())),
        **{"in::" + k: v.numpy() for k, v in batch.items() if k != "pixel_values"},
        mask_logits=mask_logits.detach().numpy(), trans_rows=trans.detach()[ar[:, None], rows].numpy(), trans_row_index=rows.numpy(),
        loss=np.float64(float(loss.detach())), ranks=np.asarray(ranks), none_grad=np.array(none_grad),
        grad_norm_names=np.array(sorted(norms)), grad_norm_vals=np.array([norms[k] for k in sorted(norms)]), **samples, **tap_out)
    print(f"g9b_flava_b8: loss {float(loss):.6f} ranks {np.asarray(ranks).tolist()} taps {sorted(taps)} none-grad {len(none_grad)} ({time.time() - t0:.1f} s)", flush=True)


def main():
    p = ap.ArgumentParser()
    p.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    p.add_argument("--only", default="")
    a = p.parse_args()
    torch.set_num_threads(os.cpu_count() or 8)
    lit, unimo = load_reference()
    jobs = [
        dict(tag="g7_bench_plain", patch=16, B=32, L=64, pretrain=False, conditioned=False, weight_seed=0, batch_seed=1234, batch_total=256, taps=True),
        dict(tag="g7_bench_cond", patch=16, B=32, L=64, pretrain=False, conditioned=True, weight_seed=0, batch_seed=1234, batch_total=256, taps=False),
        dict(tag="g8_pretrain_cond", patch=32, B=8, L=96, pretrain=True, conditioned=True, weight_seed=0, batch_seed=1234, batch_total=8, taps=False),
        dict(tag="g8_pretrain_plain", patch=32, B=8, L=96, pretrain=True, conditioned=False, weight_seed=0, batch_seed=1234, batch_total=8, taps=False),
        # G8b (round 3): the pre-train step at the ViT-B/16 geometry of BASELINE configs[1]/[4] (P=196, 393 vision tokens)
        dict(tag="g8b_pretrain_p196_cond", patch=16, B=8, L=96, pretrain=True, conditioned=True, weight_seed=0, batch_seed=1234, batch_total=8, taps=False),
    ]
    for j in jobs:
        if a.only and a.only not in j["tag"]:
            continue
        run(lit, unimo, a.out, **j)
    if not a.only or a.only in "g9_flava_real":
        g9_flava(a.out)
    if not a.only or a.only in "g9b_flava_b8":
        g9b_flava(a.out)


if __name__ == "__main__":
    main()
