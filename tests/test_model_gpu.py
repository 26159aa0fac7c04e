"""End-to-end parity of the HIP MKGformer path (bf16 compute) against the CPU oracle (fp32), real dimensions.
Tolerances follow BASELINE.json north_star: logits within 1e-2 (bf16); ranks exact where the margin exceeds it."""
import argparse
import os
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mkgformer_oracle as O  # noqa: E402  (tests may use the oracle; the product never does)

BASE, NE, NR = 30522, 11292, 192


def _condition(sd):
    """Well-conditioned variant of the synthetic weights.  With plain N(0,0.02) weights the UNSCALED fusion softmax
    softmax(ctx vis^T) of text layers 8-11 (modeling_unimo.py:405-410) is nearly one-hot (score std ~15), so the
    network is chaotic there: rounding only the weight matrices to bf16 inside the fp32 oracle already moves
    trans_hidden by ~5% and the logits by >0.1 (see test_bf16_sensitivity_control).  Shrinking the text value
    projection of those layers keeps every code path live but makes the map smooth, so implementation parity can be
    asserted at the 1e-2 logit tolerance of BASELINE.json."""
    for l in range(8, 12):
        for k in ("weight", "bias"):
            n = f"unimo.encoder.text_layer.{l}.attention.self.value.{k}"
            sd[n] = sd[n] * 0.05
    return sd


def _oracle_sd(vc, seed, analogy_rel, conditioned=False):
    sd = O.init_params(vc, O.TextCfg(vocab_size=BASE + NE + NR), seed=seed)
    if conditioned:
        sd = _condition(sd)
    return O.init_relation_word(sd, analogy_rel)


def _product(vc_patch, seed, conditioned=False):
    from mkg_analogy_amd import data_synth as D
    from mkg_analogy_amd.lit_models import TransformerLitModel
    from mkg_analogy_amd.models import MKGformerKGC, TextConfig, VisionConfig
    torch.manual_seed(0)
    model = MKGformerKGC(VisionConfig(patch_size=vc_patch), TextConfig())
    cfg = D.data_config(seed=1234)
    args = argparse.Namespace(label_smoothing=0.1, alpha=0.43, pretrain=0, lr=5e-5, weight_decay=0.01, optimizer="AdamW", warm_up_radio=0.1)
    tok = D.FakeTokenizer()
    lit = TransformerLitModel(model=model, args=args, tokenizer=tok, data_config=cfg)          # resize -> 42006
    vc = O.VisionCfg(patch_size=vc_patch)
    sd0 = O.init_params(vc, O.TextCfg(vocab_size=BASE + NE + NR), seed=seed)
    if conditioned:
        sd0 = _condition(sd0)
    missing, unexpected = model.load_state_dict(sd0, strict=False)
    assert not unexpected and all(("position_ids" in m or "decoder" in m) for m in missing), (missing, unexpected)
    model.cuda()
    lit._init_relation_word()                                                                  # -> 42007, [R] = mean of relation rows
    return model, lit, cfg, vc


def _stats(name, got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    err = (got - ref).abs().max().item()
    rel = ((got - ref).norm() / (ref.norm() + 1e-30)).item()
    print(f"   {name}: max|err| {err:.3e}  rel-L2 {rel:.3e}  ref max {ref.abs().max().item():.3e}")
    return err, rel


@pytest.mark.parametrize("patch,B,conditioned,L", [(32, 4, True, 64), (16, 2, True, 64), (32, 4, False, 64),
                                                    (32, 3, True, 37), (32, 1, True, 40), (32, 2, True, 160)])    # ragged: odd batch, L no multiple of 8 / 32 (general fusion and attention paths); L = 160: the long-stream two-pass text attention backward
def test_forward_backward_vs_oracle(patch, B, conditioned, L):
    from mkg_analogy_amd import data_synth as D
    model, lit, cfg, vc = _product(patch, seed=3, conditioned=conditioned)
    sd = _oracle_sd(vc, 3, cfg["analogy_relation_ids"], conditioned)
    tc = O.TextCfg(vocab_size=BASE + NE + NR + 1)
    # [R] row parity with the reference semantics
    w = model.get_input_embeddings().weight
    assert w.shape[0] == D.VOCAB
    np.testing.assert_allclose(w[-1].detach().cpu().numpy(), sd["unimo.text_embeddings.word_embeddings.weight"][-1].numpy(), atol=1e-6)

    batch = D.make_batch(B, L, seed=11)
    ids = torch.tensor(cfg["analogy_entity_ids"])
    # ---------------- oracle (fp32, CPU)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    _, trans_ref = O.forward(sdg, vc, tc, batch["input_ids"], batch["attention_mask"], batch["token_type_ids"], batch["pixel_values"],
                             batch["sep_idx"], train=False)
    loss_ref, ml_ref = O.finetune_loss(sdg, trans_ref, batch["input_ids"], batch["label"], batch["rel_idx"], batch["q_head_idx"],
                                       batch["a_head_idx"], ids, alpha=0.43)
    loss_ref.backward()
    # ---------------- HIP path
    model.eval()
    gb = {k: v.cuda() for k, v in batch.items()}
    st = model.store
    st.zero_grad()
    loss = lit.training_step(dict(gb), 1)
    loss.backward()
    torch.cuda.synchronize()
    out, trans = model(input_ids=gb["input_ids"], attention_mask=gb["attention_mask"], token_type_ids=gb["token_type_ids"],
                       pixel_values=gb["pixel_values"], sep_idx=gb["sep_idx"], return_dict=True)
    _, mi = (gb["input_ids"] == 103).nonzero(as_tuple=True)
    ml = out.logits[torch.arange(B, device="cuda"), mi][:, ids.cuda()]
    print(f"\npatch {patch} B {B}: loss hip {float(loss):.6f} oracle {float(loss_ref):.6f}")
    e_t, r_t = _stats("trans_hidden", trans, trans_ref)
    e_l, r_l = _stats("mask logits", ml, ml_ref)
    if conditioned:
        # bf16 tolerance of BASELINE.json (1e-2 on logits), taken relative to the logit scale: a bf16 pipeline carries
        # ~2^-8 relative error per rounding, so an absolute 1e-2 is only meaningful for O(1) logits
        scale = max(1.0, float(ml_ref.detach().abs().max()))
        rms = float((ml.detach().float().cpu() - ml_ref.detach()).pow(2).mean().sqrt())
        print(f"   logit tolerance {1e-2 * scale:.3e} (scale {scale:.2f}); rms err {rms:.3e}")
        assert e_l < 1e-2 * scale, "mask logits outside the bf16 tolerance (1e-2 x logit scale) of BASELINE.json"
        assert rms < 5e-3
        assert r_t < 1.5e-2
        assert abs(float(loss) - float(loss_ref)) < 6e-3      # mean over B<=4 rows of -0.9*logit[label] + ...: inherits the logit error
        gtol = 0.08
    else:
        # chaotic regime: compare with the intrinsic bf16 sensitivity of the reference math itself
        sdb = {k: (v.detach().to(torch.bfloat16).float() if v.dim() >= 2 and "embeddings" not in k else v.detach()) for k, v in sd.items()}
        with torch.no_grad():
            _, trans_ctl = O.forward(sdb, vc, tc, batch["input_ids"], batch["attention_mask"], batch["token_type_ids"], batch["pixel_values"],
                                     batch["sep_idx"], train=False)
            loss_ctl, _ = O.finetune_loss(sdb, trans_ctl, batch["input_ids"], batch["label"], batch["rel_idx"], batch["q_head_idx"],
                                          batch["a_head_idx"], ids, alpha=0.43)
        ctl = ((trans_ctl - trans_ref.detach()).norm() / trans_ref.detach().norm()).item()
        lctl = abs(float(loss_ctl) - float(loss_ref))
        print(f"   control (fp32 oracle with bf16-rounded weights only): trans rel-L2 {ctl:.3e}, loss moves by {lctl:.3e}")
        assert r_t < 3.0 * ctl + 1e-2, "drift exceeds the intrinsic bf16 sensitivity of the reference math"
        # the loss inherits the same sensitivity (two correct bf16 implementations that differ by one rounding land 0.01-0.1
        # apart here): it is held to the control's own displacement, not to an absolute number
        assert abs(float(loss) - float(loss_ref)) < 3.0 * lctl + 2e-2
        gtol = 0.6
    # ranks: exact wherever the label's margin to every other class exceeds the logit tolerance
    ranks_ref = O.ranks_double_sort(ml_ref.detach(), batch["label"])
    ev = lit._eval(dict(gb), 0)
    amb = ((ml_ref.detach() - ml_ref.detach()[torch.arange(B), batch["label"]][:, None]).abs() < 2 * e_l).sum(1).numpy() - 1
    print("   ranks hip", ev["entity_ranks"], "oracle", ranks_ref, "ambiguous", amb)
    assert np.all(np.abs(ev["entity_ranks"] - ranks_ref) <= amb)
    # gradients
    worst = 0.0
    names = ["cls.predictions.transform.dense.weight", "cls.predictions.bias", "unimo.text_embeddings.word_embeddings.weight",
             "unimo.text_embeddings.position_embeddings.weight", "unimo.text_embeddings.LayerNorm.weight",
             "unimo.vision_embeddings.patch_embedding.weight", "unimo.vision_embeddings.class_embedding",
             "unimo.vision_embeddings.position_embedding.weight", "unimo.vision_pre_layrnorm.bias"]
    for l in (0, 5, 7, 8, 11):
        t, v = f"unimo.encoder.text_layer.{l}.", f"unimo.encoder.vision_layers.{l}."
        names += [t + "attention.self.query.weight", t + "attention.self.key.bias", t + "attention.self.value.weight",
                  t + "attention.self.adaptive_weight.0", t + "attention.self.adaptive_weight.1", t + "attention.output.dense.weight",
                  t + "attention.output.LayerNorm.weight", t + "intermediate.dense.weight", t + "intermediate.dense.bias",
                  t + "output.dense.weight", t + "output.LayerNorm.bias",
                  v + "self_attn.q_proj.weight", v + "self_attn.k_proj.bias", v + "self_attn.v_proj.weight", v + "self_attn.out_proj.weight",
                  v + "layer_norm1.weight", v + "mlp.fc1.weight", v + "mlp.fc1.bias", v + "mlp.fc2.weight", v + "layer_norm2.bias"]
        if l >= 8:
            names += [t + "intermediate.fusion_dense.weight", t + "intermediate.fusion_dense.bias"]
    for n in names:
        g, r = st.g(n).detach().float().cpu(), sdg[n].grad
        if r.norm().item() < 1e-7:            # mathematically zero (e.g. CLIP k_proj.bias: softmax is shift invariant over keys)
            assert g.norm().item() < 1e-3, n
            continue
        rel = ((g - r).norm() / (r.norm() + 1e-20)).item()
        cos = torch.nn.functional.cosine_similarity(g.flatten(), r.flatten(), dim=0).item()
        worst = max(worst, rel)
        flag = "" if rel < gtol else "   <-- CHECK"
        print(f"   grad {n}: rel-L2 {rel:.3e} cos {cos:.5f} |ref| {r.norm().item():.3e}{flag}")
        if conditioned and g.numel() == 1:
            # scalar parameters (the adaptive attention weights): the gradient is ONE sum of B * heads * L * L signed terms, and with B = 1 it
            # can cancel to 2e-5 where its neighbours are 5e-3 -- the bf16 noise of the sum (1e-5 .. 4e-5 absolute at every layer and batch
            # size) is then a large fraction of the value; cosine similarity says nothing for one number.  Relative gate + that noise floor.
            assert abs(float(g) - float(r)) < 0.12 * abs(float(r)) + 5e-5, (n, float(g), float(r))
        elif conditioned:
            assert cos > 0.99 and rel < 0.12, n
        # (chaotic regime: gradients are printed for information only -- the map itself is not Lipschitz-stable under bf16)
    # tensors that get no gradient in the reference stay at zero
    for n in ("unimo.text_pooler.dense.weight", "unimo.vision_post_layernorm.weight"):
        assert float(st.g(n).abs().max()) == 0.0
    # fusion_dense of layers < 8 is unused -> zero gradient (reference: None)
    assert float(st.g("unimo.encoder.text_layer.3.intermediate.fusion_dense.weight").abs().max()) == 0.0


def test_train_mode_step_and_optimizer():
    """Dropout on: loss finite and close to eval loss; 3 fused-AdamW steps reduce the loss on a fixed batch;
    parameter update matches torch.optim.AdamW applied to the same gradients."""
    from mkg_analogy_amd import data_synth as D
    from mkg_analogy_amd.trainer import Trainer
    model, lit, cfg, vc = _product(32, seed=5)
    gb = D.make_batch(8, 64, seed=21, device="cuda")
    tr = Trainer(max_epochs=1, max_steps=40)
    tr._setup(lit, [gb] * 40)
    st = model.store
    name = "unimo.encoder.vision_layers.3.mlp.fc1.weight"
    losses = []
    for i in range(4):
        before = st.m(name).clone()
        lr = tr.optimizer.param_groups[0]["lr"]
        loss = tr.train_step(lit, gb, i)
        losses.append(float(loss))
        g = st.g(name).clone()
        if i == 1:          # step 2 (lr > 0): check one tensor against torch.optim.AdamW given the same grad history is impractical;
            # instead verify the closed form for step count i+1 using the optimizer's own moments
            m, v = tr.optimizer.m[st.slots[name].offset:st.slots[name].offset + g.numel()].view_as(g), \
                tr.optimizer.v[st.slots[name].offset:st.slots[name].offset + g.numel()].view_as(g)
            stepn = tr.optimizer.steps
            ref = before * (1 - lr * 0.01) - (lr / (1 - 0.9 ** stepn)) * m / (v.sqrt() / math.sqrt(1 - 0.999 ** stepn) + 1e-8)
            assert torch.allclose(st.m(name), ref, atol=1e-7, rtol=1e-5)
            assert torch.allclose(st.w(name).float(), st.m(name), atol=1e-2, rtol=1e-2)
            assert torch.equal(st.wt("v3.fc1"), st.w(name).t())
    print("train-mode losses", losses)
    assert all(math.isfinite(x) for x in losses)
    assert losses[-1] < losses[0]
    m = tr.validate(lit, [gb])
    assert "Eval_entity/hits1" in m and 0 <= m["Eval_entity/hits1"] <= 1


def test_pretrain_branch():
    from mkg_analogy_amd import data_synth as D
    model, lit, cfg, vc = _product(32, seed=7, conditioned=True)
    lit.args.pretrain = 1
    batch = D.make_batch(6, 96, seed=31, pretrain=True)
    sd = _oracle_sd(vc, 7, cfg["analogy_relation_ids"], conditioned=True)
    tc = O.TextCfg(vocab_size=BASE + NE + NR + 1)
    _, trans_ref = O.forward(sd, vc, tc, batch["input_ids"], batch["attention_mask"], batch["token_type_ids"], batch["pixel_values"], None, train=False)
    loss_ref = O.pretrain_loss(sd, trans_ref, batch["input_ids"], batch["label"], batch["pre_type"], (BASE, BASE + NE), (BASE + NE, BASE + NE + NR))
    model.eval()
    gb = {k: v.cuda() for k, v in batch.items()}
    model.store.zero_grad()
    loss = lit.training_step(dict(gb), 1)
    loss.backward()
    print("pretrain loss hip", float(loss), "oracle", float(loss_ref))
    assert abs(float(loss) - float(loss_ref)) < 1e-2
    ev = lit._eval(dict(gb), 0)
    assert "entity_ranks" in ev or "relation_ranks" in ev


def test_image_index_path_equals_pixel_values_path():
    """Device-side batch assembly: indices into the resident image table give bit-identical outputs
    to the stacked pixel_values the reference collator would have built."""
    from mkg_analogy_amd import data_synth as D
    from mkg_analogy_amd.batching import DeviceImageTable
    model, lit, cfg, vc = _product(32, seed=9, conditioned=True)
    model.eval()
    B = 4
    batch = D.make_batch(B, 64, seed=41, device="cuda")
    names = [f"Q{i}" for i in range(6)]
    table = torch.randn(6, 3, 224, 224)
    tab = DeviceImageTable(names)
    head = ["Q1", "Q3", None, "Q5"]
    tail = ["Q2", None, None, "Q0"]
    idx = tab.slots(head, tail)
    pix = torch.zeros(B, 2, 3, 224, 224)
    for b in range(B):
        for s in range(2):
            if idx[b, s] >= 0:
                pix[b, s] = table[idx[b, s]]
    model.set_image_table(table)
    kw = dict(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], token_type_ids=batch["token_type_ids"],
              sep_idx=batch["sep_idx"], return_dict=True)
    with torch.no_grad():
        _, t1 = model(pixel_values=pix.cuda(), **kw)
        _, t2 = model(image_index=idx, **kw)
    assert torch.equal(t1, t2)


def test_full_size_training_steps_stay_finite():
    """BASELINE configs[1] size (B=256, 393 vision tokens, L=64), 6 optimizer steps: loss, gradients and weights stay
    finite.  Regression for a barrier bug in the attention tile loops (waves whose query rows lie past the end skipped the
    compute body and with it the only wait on their share of the LDS-DMA) that showed up as rare NaNs only at this size."""
    import bench
    from mkg_analogy_amd import data_synth as D
    from mkg_analogy_amd.trainer import Trainer
    dev = torch.device("cuda", 0)
    model, lit, cfg = bench.build(16, seed=0, device=dev, backbone="mkgformer")
    batch = D.make_batch(256, 64, seed=1234, device=dev)
    tr = Trainer(max_epochs=1, max_steps=150, world_size=1)
    tr._setup(lit, [None] * 150)
    losses = []
    for i in range(6):
        loss = tr.train_step(lit, batch, i)
        torch.cuda.synchronize()
        losses.append(float(loss))
        assert math.isfinite(losses[-1]), losses
        assert bool(torch.isfinite(model.store.grad).all()) and bool(torch.isfinite(model.store.master).all()), f"step {i}"
    print("\nfull-size losses", [round(x, 4) for x in losses])
    assert losses[-1] < losses[0]


def test_full_size_forward_is_per_example():
    """Size-independent properties at BASELINE configs[1] size (B=256, 393 vision tokens, L=64, eval mode): every example
    is independent of its position in the batch and of its batch mates -- permuting the batch permutes the outputs
    BIT-EXACTLY (no tile, workgroup or stream boundary leaks rows into each other), a half batch reproduces its rows, the
    two evaluation passes of one batch are identical (no race), and ranks computed on the device agree with the double
    sort of the same logits on the host."""
    import bench
    from mkg_analogy_amd import data_synth as D
    dev = torch.device("cuda", 0)
    model, lit, cfg = bench.build(16, seed=0, device=dev, backbone="mkgformer")
    model.eval()
    batch = D.make_batch(256, 64, seed=4321, device=dev)
    keys = ("input_ids", "attention_mask", "token_type_ids", "pixel_values", "sep_idx")

    def fwd(bt):
        with torch.no_grad():
            out, trans = model(**{k: bt[k] for k in keys}, return_dict=True)
        return trans.float().clone()
    t0 = fwd(batch)
    assert bool(torch.isfinite(t0).all())
    assert torch.equal(t0, fwd(batch)), "two passes over the same batch differ"
    perm = torch.randperm(256, generator=torch.Generator().manual_seed(1)).to(dev)
    tp = fwd({k: batch[k][perm] for k in keys})
    assert torch.equal(tp, t0[perm]), "outputs depend on the position of an example in the batch"
    th = fwd({k: batch[k][:128] for k in keys})
    err = float((th - t0[:128]).abs().max())
    print("\nhalf batch vs full batch max|diff|", err)
    assert err <= 2e-2 * float(t0.abs().max())                    # other tile shapes may be picked: same math, rounding-level differences
    # ranks: device double-sort == host double-sort on the very same logits (bit-exact integer work)
    ev = lit._eval({k: v for k, v in batch.items()}, 0)
    ranks = ev["entity_ranks"] if isinstance(ev, dict) else ev
    with torch.no_grad():
        out, _ = model(**{k: batch[k] for k in keys}, return_dict=True)
        mask_idx = (batch["input_ids"] == 103).nonzero()[:, 1]
        ids = torch.tensor(cfg["analogy_entity_ids"], device=dev)
        logits = out.logits[torch.arange(256, device=dev), mask_idx][:, ids].float().cpu()
    order = torch.argsort(logits, dim=1, descending=True, stable=True)
    host = (torch.argsort(order, dim=1, stable=True)[torch.arange(256), batch["label"].cpu()] + 1).numpy()
    import numpy as np
    # torch.sort is not stable in the reference, so only rows whose label logit is not tied with another are compared
    lab = logits[torch.arange(256), batch["label"].cpu()]
    untied = ((logits == lab[:, None]).sum(1) == 1).numpy()
    assert untied.sum() > 200
    assert np.array_equal(np.asarray(ranks)[untied], host[untied])


def test_full_size_gradient_of_duplicated_batch():
    """Backward at BASELINE configs[1] size through a size-independent property: the mean-loss gradient of a batch made of
    two identical halves (B=256) equals the gradient of one half (B=128) -- every M-split reduction, atomic accumulation
    and stream join of the backward pass at full size, checked without an oracle run (which takes minutes at this size).
    Power-of-two loss scaling keeps the bf16 roundings aligned, so only fp32 summation order differs."""
    import bench
    from mkg_analogy_amd import data_synth as D
    dev = torch.device("cuda", 0)
    model, lit, cfg = bench.build(16, seed=0, device=dev, backbone="mkgformer")
    model.eval()                                                   # dropout off: the two halves see the same network
    half = D.make_batch(128, 64, seed=777, device=dev)
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
    print(f"\nloss half {l1:.6f} full {l2:.6f}; gradient rel-L2 difference {rel:.2e}; |g| {float(g1.norm()):.4f}")
    assert abs(l1 - l2) < 1e-5 * max(1.0, abs(l1))
    assert rel < 1e-3


def test_unsynchronised_steps_do_not_grow_memory():
    """A training loop that never synchronises (what bench.py times) must not let the host run arbitrarily far ahead:
    blocks touched by the side streams only return to the caching allocator once those streams pass the free, so an
    unbounded run-ahead reserved ~2.3 GiB more per queued step at B=256 until the device was full and every allocation
    became free-everything-and-retry (100 -> 600 ms per step after ~60 steps).  The engine bounds the passes in flight."""
    import bench
    from mkg_analogy_amd import data_synth as D
    from mkg_analogy_amd.trainer import Trainer
    dev = torch.device("cuda", 0)
    model, lit, cfg = bench.build(32, seed=0, device=dev, backbone="mkgformer")
    batch = D.make_batch(64, 64, seed=99, device=dev)
    tr = Trainer(max_epochs=1, max_steps=1000, world_size=1)
    tr._setup(lit, [None] * 1000)
    for i in range(6):
        tr.train_step(lit, batch, i)
    torch.cuda.synchronize()
    r0 = torch.cuda.memory_reserved()
    for i in range(60):
        tr.train_step(lit, batch, 6 + i)
    torch.cuda.synchronize()
    r1 = torch.cuda.memory_reserved()
    print(f"\nreserved after 6 steps {r0 / 2**30:.2f} GiB, after 66 steps {r1 / 2**30:.2f} GiB")
    assert r1 <= 1.5 * r0 + (1 << 30)


def test_pool_headroom_stops_allocator_growth():
    """Trainer._pool_headroom: after the third step the caching allocator's pool holds what two steps in flight need, on every stream pool -- the steps
    that follow make (next to) no device allocation, also when the first three ran with the host NOT ahead of the GPU (a cold process; emulated with a
    synchronize after each), the case in which the whole second in-flight step's buffers (~140 allocations at B = 256) used to be allocated later, inside
    whatever steps were being timed.  Without the headroom (MART_POOL_HEADROOM=0) the same loop allocates several dozen times."""
    import bench
    from mkg_analogy_amd import data_synth as D
    from mkg_analogy_amd.trainer import Trainer
    dev = torch.device("cuda", 0)

    def run(headroom):
        old = os.environ.get("MART_POOL_HEADROOM")
        os.environ["MART_POOL_HEADROOM"] = headroom
        try:
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            torch.cuda.reset_peak_memory_stats()
            model, lit, cfg = bench.build(32, seed=0, device=dev, backbone="mkgformer")
            batch = D.make_batch(256, 64, seed=99, device=dev)      # (B = 256: a 33 ms step, so that the host does run ahead of the GPU)
            tr = Trainer(max_epochs=1, max_steps=1000, world_size=1)
            tr._setup(lit, [None] * 1000)
            for i in range(3):
                tr.train_step(lit, batch, i)
                torch.cuda.synchronize()
            n0 = torch.cuda.memory_stats()["num_device_alloc"]
            for i in range(10):
                tr.train_step(lit, batch, 3 + i)
            torch.cuda.synchronize()
            return torch.cuda.memory_stats()["num_device_alloc"] - n0
        finally:
            model = lit = tr = batch = None
            import gc
            gc.collect()
            if old is None:
                os.environ.pop("MART_POOL_HEADROOM", None)
            else:
                os.environ["MART_POOL_HEADROOM"] = old

    without, with_ = run("0"), run("0.10")
    print(f"\ndevice allocations in 10 un-synchronised steps after a cold start: {without} without the headroom, {with_} with it")
    assert with_ <= 3, with_
    assert without >= 20, without          # (the situation the headroom exists for is the one this test sets up)


def test_streamed_adamw_equals_one_shot():
    """The optimizer step released range by range (as the backward pass frees gradients) == the one-launch step on the same
    gradients, bit for bit (same kernel, same chunk table, only the launch partition differs)."""
    import copy
    from mkg_analogy_amd import data_synth as D
    from mkg_analogy_amd.optim import FusedAdamW
    model, lit, cfg, vc = _product(32, seed=11, conditioned=True)
    st = model.store
    g = torch.Generator(device="cpu").manual_seed(3)
    st.grad.copy_((torch.randn(st.total, generator=g) * 1e-3).cuda())
    m0 = st.master.clone()
    opt_a = FusedAdamW(model, lr=1e-3)
    for _ in range(2):
        opt_a.step()
    ref = st.master.clone(); ref_m = opt_a.m.clone(); ref_sh = st.shadow.clone()
    st.master.copy_(m0)
    st.refresh_shadows()
    opt_b = FusedAdamW(model, lr=1e-3)
    for _ in range(2):
        opt_b.begin_step()
        for frac in (0.1, 0.1, 0.37, 0.9):                      # repeated / growing offsets, not aligned to chunks
            opt_b.ready(int(st.total * frac))
        opt_b.step()
    torch.cuda.synchronize()
    assert torch.equal(st.master, ref) and torch.equal(opt_b.m, ref_m) and torch.equal(st.shadow, ref_sh)
    assert opt_a.steps == opt_b.steps == 2


def test_gradients_are_run_to_run_identical():
    """Two backward passes from the same weights on the same batch give BIT-IDENTICAL gradients for every tensor: GEMM weights and
    biases (gemm_tn: partial tiles through a workspace, summed in split order), LayerNorm weights / biases (per-workgroup
    partials, summed in workgroup order), the adaptive weights (per-wave slots + one reducing workgroup), the tied decoder rows /
    bias, the word-embedding rows (summed in sorted-token order) and the position / type / class rows (slice partials, ordered
    reduction).  No float atomics are left on the gradient path (MART_DETERMINISTIC=0 restores them)."""
    from mkg_analogy_amd import data_synth as D
    model, lit, cfg, vc = _product(32, seed=13, conditioned=True)
    gb = D.make_batch(16, 64, seed=51, device="cuda")
    st = model.store
    for mode in ("eval", "train"):
        getattr(model, mode)()
        grads = []
        for _ in range(2):
            model._step = 7                                            # same dropout masks in both passes
            st.zero_grad()
            loss = lit.training_step(dict(gb), 1)
            loss.backward()
            torch.cuda.synchronize()
            grads.append((float(loss.detach()), st.grad.clone()))
        assert grads[0][0] == grads[1][0], mode
        differ = [n for n, sl in st.slots.items()
                  if not torch.equal(grads[0][1][sl.offset:sl.offset + sl.numel], grads[1][1][sl.offset:sl.offset + sl.numel])]
        assert not differ, (mode, differ[:10])


def test_changing_batch_shapes_leave_no_stale_state():
    """The reference pads every batch to its own longest example (data_module.py:113-119), so consecutive steps see different (B, L).  A pass on
    one shape, then passes on others, then the first shape again: loss, [MASK] logits and every gradient of the repeated pass are BIT-IDENTICAL
    to the first one, in eval and in train mode (same dropout step) -- nothing cached per shape (row bookkeeping, [MASK] positions, operand
    twins, kernel plans, workspaces) survives into a pass it does not belong to.  And a model that has only ever seen the second shape agrees
    with the one that saw it between two others."""
    from mkg_analogy_amd import data_synth as D
    model, lit, cfg, vc = _product(32, seed=13, conditioned=True)
    fresh, lit_f, _, _ = _product(32, seed=13, conditioned=True)
    shapes = [(8, 64), (5, 37), (8, 57), (3, 44), (8, 64), (5, 37)]
    batches = {sh: D.make_batch(sh[0], sh[1], seed=60 + sh[1], device="cuda") for sh in set(shapes)}
    st = model.store

    def one(m, l, gb, mode):
        getattr(m, mode)()
        m._step = 7
        m.store.zero_grad()
        loss = l.training_step(dict(gb), 1)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach()), m.store.grad.clone()

    for mode in ("eval", "train"):
        seen = {}
        for sh in shapes:
            got = one(model, lit, batches[sh], mode)
            assert np.isfinite(got[0]), (mode, sh)
            if sh in seen:
                assert got[0] == seen[sh][0], (mode, sh, got[0], seen[sh][0])
                differ = [n for n, sl in st.slots.items()
                          if not torch.equal(got[1][sl.offset:sl.offset + sl.numel], seen[sh][1][sl.offset:sl.offset + sl.numel])]
                assert not differ, (mode, sh, differ[:10])
            seen[sh] = got
        ref = one(fresh, lit_f, batches[(5, 37)], mode)
        assert ref[0] == seen[(5, 37)][0], (mode, ref[0], seen[(5, 37)][0])
        assert torch.equal(ref[1], seen[(5, 37)][1]), mode


def test_fused_fusion_kernels_equal_the_general_path_in_the_model():
    """BertFusion through mart_fusion_fwd / mart_fusion_bwd (one kernel per direction) against the GEMM / softmax / transpose launches
    they replace, inside the full model (P=196: 393 vision tokens, L=64): same logits and gradients up to the bf16 rounding of
    the probabilities -- both paths round them to bf16 once, the fused one sums d(visual) in a different order."""
    from mkg_analogy_amd import data_synth as D
    model, lit, cfg, vc = _product(16, seed=5, conditioned=True)
    B, L = 8, 64
    batch = D.make_batch(B, L, seed=21, device="cuda")
    lit.train(); model.eval()
    eng = model.engine
    assert eng.fused_fusion

    st = model.store

    def run(fused):
        eng.fused_fusion = fused
        st.zero_grad()
        loss = lit.training_step(dict(batch), 0)
        loss.backward()
        torch.cuda.synchronize()
        flat = st.grad.clone()
        return float(loss.detach()), {n: flat[sl.offset:sl.offset + sl.numel] for n, sl in st.slots.items()}

    l1, g1 = run(True)
    l0, g0 = run(False)
    eng.fused_fusion = True
    print(f"\nloss fused {l1:.6f} general {l0:.6f}")
    assert abs(l1 - l0) < 2e-3
    worst, compared, aw0, aw1 = 0.0, 0, [], []
    for n in g0:
        d0 = float(g0[n].norm())
        if d0 < 1e-8 or n.endswith(("k_proj.bias", "key.bias")):   # key biases: zero gradient in exact arithmetic (softmax shift invariance), rounding noise here
            continue
        compared += 1
        r = float((g1[n] - g0[n]).norm()) / d0
        if "adaptive_weight" in n:                                 # scalars: sums of 10^5 signed terms each; held as ONE vector below
            aw0.append(g0[n].reshape(-1)); aw1.append(g1[n].reshape(-1))
            continue
        worst = max(worst, r)
        assert r < 3e-2, (n, r)
    aw0, aw1 = torch.cat(aw0), torch.cat(aw1)
    assert float((aw1 - aw0).norm() / aw0.norm()) < 0.15
    assert compared > 400 and worst > 0.0                        # all gradient tensors took part, and the two paths really are different code
    print(f"worst relative gradient difference fused vs general: {worst:.3e} over {compared} tensors")


def test_async_step_boundary_equals_in_order():
    """optim.FusedAdamW.async_step (MART_ASYNC_STEP=1): the gradient zero-fill and the W^T refresh on the optimizer stream, joined by the model's
    forward / FlatStore.g() / wt().  Three training steps (eval-mode arithmetic: no dropout) end in bit-identical weights, moments and shadows."""
    from mkg_analogy_amd import data_synth as D
    from mkg_analogy_amd.trainer import Trainer
    gb = D.make_batch(8, 64, seed=21, device="cuda")
    res = []
    for mode in (False, True):
        model, lit, cfg, vc = _product(32, seed=5)
        tr = Trainer(max_epochs=1, max_steps=40)
        tr._setup(lit, [gb] * 40)
        tr.optimizer.async_step = mode
        model.engine.p_hidden = model.engine.p_attn = 0.0          # same arithmetic in both runs
        for i in range(3):
            tr.train_step(lit, gb, i)
        torch.cuda.synchronize()
        st = model.store
        res.append((st.master.clone(), st.shadow.clone(), st.shadow_t.clone(), tr.optimizer.m.clone(), tr.optimizer.v.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_store_ownership_fast_path_and_rebuild():
    """``model.store`` / ``model.engine`` are touched eight times per step: the ownership check is FlatStore.still_bound() (dict lookups + addresses),
    the named_parameters() walk only runs when that fails -- and it does fail, and the store is rebuilt, when the parameters move (.cpu() / .cuda())
    or a Parameter object is re-assigned."""
    from mkg_analogy_amd import data_synth as D
    model, lit, cfg, vc = _product(32, seed=5, conditioned=True)
    st = model.store
    assert st.still_bound() and model.store is st and model.engine is model.engine
    gb = {k: v.cuda() for k, v in D.make_batch(2, 64, seed=3).items()}
    model.eval()
    kw = dict(input_ids=gb["input_ids"], attention_mask=gb["attention_mask"], token_type_ids=gb["token_type_ids"], pixel_values=gb["pixel_values"],
              sep_idx=gb["sep_idx"], return_dict=True)
    with torch.no_grad():
        _, t0 = model(**kw)
    model.cpu(); model.cuda()                                     # every parameter leaves the flat buffer
    assert not st.still_bound()
    st2 = model.store
    assert st2 is not st and st2.still_bound()
    with torch.no_grad():
        _, t1 = model(**kw)
    assert torch.equal(t0, t1)
    b = model.cls.predictions.transform.LayerNorm.bias
    model.cls.predictions.transform.LayerNorm.bias = torch.nn.Parameter(b.detach().clone() + 1.0)      # a re-assigned Parameter object
    assert not st2.still_bound()
    st3 = model.store
    assert st3 is not st2
    with torch.no_grad():
        _, t2 = model(**kw)
    assert not torch.equal(t1, t2)                                 # the new bias is the one the kernels read


def test_example_without_mask_token_is_flagged_not_nan():
    """an example that holds no [MASK] (truncated / malformed).  The reference raises a shape mismatch at
    ``logits[arange(bs), mask_idx]`` (lit_models/transformer.py:94-95).  Here the step stays finite -- ``needed_rows`` and the [MASK]-row lookup
    agree on row 0 of that example, so nothing NaN reaches the loss, the gradients or AdamW -- and ``Fn.check_status`` raises for it."""
    from mkg_analogy_amd import data_synth as D
    from mkg_analogy_amd import functional as Fn
    model, lit, cfg, vc = _product(32, seed=3, conditioned=True)
    assert lit.last_layer_rows
    batch = D.make_batch(4, 64, seed=11)
    ids = batch["input_ids"].clone()
    ids[2][ids[2] == 103] = 1999                        # example 2 loses its [MASK]
    batch["input_ids"] = ids
    gb = {k: v.cuda() for k, v in batch.items()}
    Fn.check_status()
    model.train()
    st = model.store
    st.zero_grad()
    loss = lit.training_step(dict(gb), 1)
    loss.backward()
    torch.cuda.synchronize()
    assert math.isfinite(float(loss))
    assert bool(torch.isfinite(st.grad).all())
    with pytest.raises(IndexError, match="MASK"):
        Fn.check_status()
    Fn.check_status()                                   # reported once
    # the evaluation pass flags it as well (Trainer._run_eval checks after the pass)
    model.eval()
    out = lit.validation_step(dict(gb), 0)
    assert out["entity_ranks"].shape == (4,)
    with pytest.raises(IndexError, match="MASK"):
        Fn.check_status()


def test_torch_optimizer_by_name_equals_fused_adamw_and_runs_adam():
    """lit_models/base.py:31 takes any torch.optim class by name.  (1) torch.optim.AdamW routed through TorchOptimizerOnStore gives the fused
    kernel's update on the same gradients (fp32 rounding), dead tensors untouched by both, shadows following the master; (2) ``--optimizer Adam``
    drives two training steps through the trainer."""
    import argparse as ap
    from mkg_analogy_amd import data_synth as D
    from mkg_analogy_amd.optim import FusedAdamW, TorchOptimizerOnStore
    from mkg_analogy_amd.trainer import Trainer
    model, lit, cfg, vc = _product(32, seed=11, conditioned=True)
    st = model.store
    g = torch.Generator(device="cpu").manual_seed(3)
    grad = (torch.randn(st.total, generator=g) * 1e-3).cuda()
    m0 = st.master.clone()
    st.grad.copy_(grad)
    fused = FusedAdamW(model, lr=1e-3)
    for _ in range(2):
        fused.step()
    ref = st.master.clone()
    st.master.copy_(m0)
    st.refresh_shadows()
    st.grad.copy_(grad)
    gen = TorchOptimizerOnStore(model, "AdamW", lr=1e-3)
    for _ in range(2):
        gen.step()
    torch.cuda.synchronize()
    assert float((st.master - ref).abs().max()) < 2e-6
    assert torch.equal(st.shadow, st.master.to(torch.bfloat16))
    s = st.slots["unimo.text_pooler.dense.weight"]                      # never receives a gradient in the reference: no update, no decay
    assert torch.equal(st.master[s.offset:s.offset + s.numel], m0[s.offset:s.offset + s.numel])
    with pytest.raises(AttributeError):
        TorchOptimizerOnStore(model, "NoSuchOptimizer")
    lit.optimizer_name = "Adam"
    tr = Trainer(max_epochs=1, max_steps=10)
    gb = {k: v.cuda() for k, v in D.make_batch(2, 64, seed=5).items()}
    tr._setup(lit, [gb] * 10)
    assert isinstance(tr.optimizer, TorchOptimizerOnStore) and type(tr.optimizer.inner).__name__ == "Adam"
    w0 = st.master.clone()
    for i in range(2):
        loss = tr.train_step(lit, gb, i)
    torch.cuda.synchronize()
    assert math.isfinite(float(loss)) and not torch.equal(st.master, w0)
    assert torch.equal(st.shadow, st.master.to(torch.bfloat16))


def test_output_hidden_states_vs_oracle():
    """``output_hidden_states=True`` (modeling_unimo.py:604-646: the text stream entering every layer + the last layer's output, 13 x [B, L, H]) on the bf16
    engine, shipped multi-queue schedule, against the oracle's per-layer text streams; the rest of the output is what a plain call returns."""
    from mkg_analogy_amd import data_synth as D
    model, lit, cfg, vc = _product(32, seed=3, conditioned=True)
    sd = _oracle_sd(vc, 3, cfg["analogy_relation_ids"], True)
    tc = O.TextCfg(vocab_size=BASE + NE + NR + 1)
    batch = D.make_batch(3, 57, seed=11)
    taps = {}
    with torch.no_grad():
        txt_emb = O.text_embed(sd, tc, batch["input_ids"], batch["token_type_ids"], False)
        _, trans_ref = O.forward(sd, vc, tc, batch["input_ids"], batch["attention_mask"], batch["token_type_ids"], batch["pixel_values"], batch["sep_idx"],
                                 train=False, taps=taps)
    gb = {k: v.cuda() for k, v in batch.items()}
    keys = ("input_ids", "attention_mask", "token_type_ids", "pixel_values", "sep_idx")
    model.eval()
    with torch.no_grad():
        out, trans = model(**{k: gb[k] for k in keys}, return_dict=True, output_hidden_states=True)
        out0, trans0 = model(**{k: gb[k] for k in keys}, return_dict=True)
        tup, _ = model(**{k: gb[k] for k in keys}, return_dict=False, output_hidden_states=True)
    hs = out.hidden_states
    assert len(hs) == 13 and all(tuple(h.shape) == (3, 57, 768) for h in hs) and out0.hidden_states is None
    assert len(tup) == 2 and len(tup[1]) == 13
    assert torch.equal(trans, trans0)
    refs = [txt_emb] + [taps[f"txt{l}"] for l in range(12)]
    worst = 0.0
    for i, (h, r) in enumerate(zip(hs, refs)):
        rel = float((h.float().cpu() - r).norm() / r.norm())
        worst = max(worst, rel)
        assert rel < (1e-5 if i == 0 else 1.5e-2), (i, rel)
    print(f"\nhidden_states: 13 text streams, worst rel-L2 vs the oracle {worst:.3e}")
    with pytest.raises(NotImplementedError):
        model(**{k: gb[k] for k in keys}, return_dict=True, output_attentions=True)
