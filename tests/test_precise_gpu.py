"""fp32-accurate evaluation path (csrc/precise.hip, engine_precise.py): north_star's fp32 gate -- logits within 1e-3 of
the fp32 CPU oracle and exact ranked indices -- at real dimensions, with the PLAIN N(0,0.02) synthetic weights (the
chaotic regime that the bf16 path cannot follow, tests/test_model_gpu.py) as well as the conditioned ones."""
import argparse

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mkgformer_oracle as O  # noqa: E402  (checker only)

DEV = "cuda"


def test_split_bf16x3_and_gemm_accuracy():
    from mkg_analogy_amd import ops
    g = torch.Generator().manual_seed(0)
    A = torch.randn(300, 768, generator=g)
    W = torch.randn(512, 768, generator=g) * 0.05
    a3 = ops.split_bf16x3(A.to(DEV), 0)
    w3 = ops.split_bf16x3(W.to(DEV), 1)
    hi = A.to(torch.bfloat16)
    lo = (A - hi.float()).to(torch.bfloat16)
    assert torch.equal(a3.cpu(), torch.cat([hi, lo, hi], 1))
    whi = W.to(torch.bfloat16)
    wlo = (W - whi.float()).to(torch.bfloat16)
    assert torch.equal(w3.cpu(), torch.cat([whi, whi, wlo], 1))
    out = torch.empty(300, 512, device=DEV)
    ops.gemm_nt(a3, w3, out)
    ref = A.double() @ W.double().T
    err = (out.cpu().double() - ref).abs().max().item()
    plain = torch.empty(300, 512, device=DEV)
    ops.gemm_nt(A.to(DEV).to(torch.bfloat16), W.to(DEV).to(torch.bfloat16), plain)
    err_bf16 = (plain.cpu().double() - ref).abs().max().item()
    print(f"\nsplit GEMM max|err| {err:.2e} (plain bf16 operands: {err_bf16:.2e}; |ref| max {ref.abs().max():.2f})")
    assert err < 2e-4 and err < err_bf16 / 50
    # strided source rows
    big = torch.randn(64, 2304, generator=g).to(DEV)
    part = ops.split_bf16x3(big[:, 768:1536], 0)
    assert torch.equal(part[:, :768].cpu(), big[:, 768:1536].cpu().to(torch.bfloat16))


def _ref_attn(q, k, v, nh, scale, pk=None, pv=None, mask=None, sep=None, w0=None, w1=None):
    B, Sq, HD = q.shape
    D = HD // nh
    sp = lambda x: x.view(B, -1, nh, D).transpose(1, 2).double()
    Q, K, V = sp(q), sp(k), sp(v)
    if pk is not None:
        K, V = torch.cat([sp(pk), K], 2), torch.cat([sp(pv), V], 2)
    s = Q @ K.transpose(-1, -2) * scale
    if sep is not None:
        for i in range(B):
            si = int(sep[i])
            s[i, :, :si, si:] *= min(max(float(w0), 0.0), 0.5)
            s[i, :, si:, si:] *= min(max(float(w1), 0.5), 1.0)
    if mask is not None:
        s = s + ((1 - mask)[:, None, None, :].double() * -10000.0)
    return (torch.softmax(s, -1) @ V).transpose(1, 2).reshape(B, Sq, HD)


def test_attn_f32_kernel():
    from mkg_analogy_amd import ops
    g = torch.Generator().manual_seed(1)
    B, nh, H, Nv, L = 3, 12, 768, 99, 40
    # vision: prefix keys, no mask
    qkv = torch.randn(B * Nv, 3 * H, generator=g)
    pre = torch.randn(B * L, 3 * H, generator=g)
    ctx = torch.empty(B * Nv, H, device=DEV)
    dq, dp = qkv.to(DEV), pre.to(DEV)
    ops.attn_fwd_f32(q=dq[:, :H], k=dq[:, H:2 * H], v=dq[:, 2 * H:], ctx=ctx, B=B, nh=nh, D=64, Sq=Nv, Sk=Nv, scale=0.125,
                     pk=dp[:, H:2 * H], pv=dp[:, 2 * H:], Lp=L)
    ref = _ref_attn(qkv[:, :H].reshape(B, Nv, H), qkv[:, H:2 * H].reshape(B, Nv, H), qkv[:, 2 * H:].reshape(B, Nv, H), nh, 0.125,
                    pre[:, H:2 * H].reshape(B, L, H), pre[:, 2 * H:].reshape(B, L, H))
    e1 = (ctx.cpu().double().view(B, Nv, H) - ref).abs().max().item()
    # text: mask + adaptive reweight
    tq = torch.randn(B * L, 3 * H, generator=g)
    mask = torch.ones(B, L, dtype=torch.int64); mask[0, 33:] = 0; mask[2, 20:] = 0
    sep = torch.tensor([[3, 5, 17, 20, 25, 30], [2, 4, 9, 12, 14, 16], [1, 2, 11, 13, 15, 19]])
    w0, w1 = torch.tensor([0.31]), torch.tensor([0.77])
    tctx = torch.empty(B * L, H, device=DEV)
    dt = tq.to(DEV)
    ops.attn_fwd_f32(q=dt[:, :H], k=dt[:, H:2 * H], v=dt[:, 2 * H:], ctx=tctx, B=B, nh=nh, D=64, Sq=L, Sk=L, scale=0.125,
                     attn_mask=mask.to(DEV), sep=sep.to(DEV)[:, 2:], sep_stride=6, w0=w0.to(DEV), w1=w1.to(DEV))
    ref = _ref_attn(tq[:, :H].reshape(B, L, H), tq[:, H:2 * H].reshape(B, L, H), tq[:, 2 * H:].reshape(B, L, H), nh, 0.125,
                    mask=mask, sep=sep[:, 2], w0=w0, w1=w1)
    e2 = (tctx.cpu().double().view(B, L, H) - ref).abs().max().item()
    # fusion: one head of 768, unscaled, sharp softmax
    c = torch.randn(B * L, H, generator=g)
    vis = torch.randn(B * Nv, H, generator=g)
    fus = torch.empty(B * L, H, device=DEV)
    ops.attn_fwd_f32(q=c.to(DEV), k=vis.to(DEV), v=vis.to(DEV), ctx=fus, B=B, nh=1, D=H, Sq=L, Sk=Nv, scale=1.0)
    ref = _ref_attn(c.view(B, L, H), vis.view(B, Nv, H), vis.view(B, Nv, H), 1, 1.0)
    e3 = (fus.cpu().double().view(B, L, H) - ref).abs().max().item()
    print(f"\nattn_f32 max|err|: vision+prefix {e1:.2e}, text mask+reweight {e2:.2e}, fusion {e3:.2e}")
    assert e1 < 2e-5 and e2 < 2e-5 and e3 < 2e-4


def test_attn_f32_mfma_kernels_at_real_shapes():
    """The f32 matrix-pipe kernels (v_mfma_f32_32x32x2_f32) at the benchmark's shapes: vision attention 393 queries x (64 prefix + 393 own) keys, the
    FLAVA text variant (row 0 exempt from the reweight), and the fusion op 64 x 393 x 768 with a sharp softmax."""
    from mkg_analogy_amd import ops
    g = torch.Generator().manual_seed(5)
    B, nh, H, Nv, L = 2, 12, 768, 393, 64
    qkv, pre = torch.randn(B * Nv, 3 * H, generator=g), torch.randn(B * L, 3 * H, generator=g)
    ctx = torch.empty(B * Nv, H, device=DEV)
    dq, dp = qkv.to(DEV), pre.to(DEV)
    ops.attn_fwd_f32(q=dq[:, :H], k=dq[:, H:2 * H], v=dq[:, 2 * H:], ctx=ctx, B=B, nh=nh, D=64, Sq=Nv, Sk=Nv, scale=0.125, pk=dp[:, H:2 * H], pv=dp[:, 2 * H:], Lp=L)
    ref = _ref_attn(qkv[:, :H].reshape(B, Nv, H), qkv[:, H:2 * H].reshape(B, Nv, H), qkv[:, 2 * H:].reshape(B, Nv, H), nh, 0.125,
                    pre[:, H:2 * H].reshape(B, L, H), pre[:, 2 * H:].reshape(B, L, H))
    e1 = (ctx.cpu().double().view(B, Nv, H) - ref).abs().max().item()
    c = 0.3 * torch.randn(B * L, H, generator=g)
    vis = torch.randn(B * Nv, H, generator=g)
    fus = torch.empty(B * L, H, device=DEV)
    ops.attn_fwd_f32(q=c.to(DEV), k=vis.to(DEV), v=vis.to(DEV), ctx=fus, B=B, nh=1, D=H, Sq=L, Sk=Nv, scale=1.0)
    ref = _ref_attn(c.view(B, L, H), vis.view(B, Nv, H), vis.view(B, Nv, H), 1, 1.0)
    e3 = (fus.cpu().double().view(B, L, H) - ref).abs().max().item()
    print(f"\nf32 MFMA kernels at real shapes: vision 393 x 457 max|err| {e1:.2e}, fusion 64 x 393 x 768 {e3:.2e}")
    assert e1 < 2e-5 and e3 < 2e-4


@pytest.mark.parametrize("Lp", [0, 64])
def test_attn_split_kernel_vs_f64_and_exact_f32(Lp):
    """mart_attn_fwd_f32 with ``fast`` (evaluation passes): the unmasked head-dim-64 attention on two-term bf16 operand splits (attn_split_fwd_k) at the
    bench shape (393 queries, 393 / 64 + 393 keys, ragged last tile), against float64 and against the exact f32 matrix-pipe kernel."""
    from mkg_analogy_amd import ops
    g = torch.Generator().manual_seed(7 + Lp)
    B, nh, H, Nv, L = 3, 12, 768, 393, 64
    qkv, pre = 1.5 * torch.randn(B * Nv, 3 * H, generator=g), 1.5 * torch.randn(B * L, 3 * H, generator=g)
    dq, dp = qkv.to(DEV), pre.to(DEV)
    kw = dict(q=dq[:, :H], k=dq[:, H:2 * H], v=dq[:, 2 * H:], B=B, nh=nh, D=64, Sq=Nv, Sk=Nv, scale=0.125,
              pk=dp[:, H:2 * H] if Lp else None, pv=dp[:, 2 * H:] if Lp else None, Lp=Lp)
    fast, exact = torch.full((B * Nv, H), float("nan"), device=DEV), torch.empty(B * Nv, H, device=DEV)
    ops.attn_fwd_f32(ctx=fast, fast=True, **kw)
    ops.attn_fwd_f32(ctx=exact, **kw)
    ref = _ref_attn(qkv[:, :H].reshape(B, Nv, H), qkv[:, H:2 * H].reshape(B, Nv, H), qkv[:, 2 * H:].reshape(B, Nv, H), nh, 0.125,
                    pre[:, H:2 * H].reshape(B, L, H) if Lp else None, pre[:, 2 * H:].reshape(B, L, H) if Lp else None)
    ef = (fast.cpu().double().view(B, Nv, H) - ref).abs().max().item()
    ee = (exact.cpu().double().view(B, Nv, H) - ref).abs().max().item()
    print(f"\nattention 393 x {Lp + Nv} keys, |out| max {ref.abs().max():.2f}: two-term splits max|err| {ef:.2e}, exact f32 kernel {ee:.2e}")
    assert bool(torch.isfinite(fast).all()) and ef < 2.5e-5 * float(ref.abs().max()) and ee < 2e-5      # 2^-16 relative for the two-term splits


def _setup(patch, seed, conditioned):
    from tests.test_model_gpu import _product, _oracle_sd
    model, lit, cfg, vc = _product(patch, seed=seed, conditioned=conditioned)
    sd = _oracle_sd(vc, seed, cfg["analogy_relation_ids"], conditioned)
    return model, lit, cfg, vc, sd


@pytest.mark.parametrize("patch,B,conditioned", [(32, 4, False), (32, 4, True), (16, 2, False)])
def test_fp32_path_logits_and_ranks(patch, B, conditioned):
    from mkg_analogy_amd import data_synth as D
    model, lit, cfg, vc, sd = _setup(patch, 3, conditioned)
    tc = O.TextCfg(vocab_size=D.VOCAB)
    batch = D.make_batch(B, 64, seed=11)
    ids = torch.tensor(cfg["analogy_entity_ids"])
    with torch.no_grad():
        _, trans_ref = O.forward(sd, vc, tc, batch["input_ids"], batch["attention_mask"], batch["token_type_ids"], batch["pixel_values"],
                                 batch["sep_idx"], train=False)
        _, mi = (batch["input_ids"] == 103).nonzero(as_tuple=True)
        ml_ref = O.score(sd, trans_ref[torch.arange(B), mi], ids)
    model.eval()
    model.set_precision("fp32")
    gb = {k: v.cuda() for k, v in batch.items()}
    out, trans = model(input_ids=gb["input_ids"], attention_mask=gb["attention_mask"], token_type_ids=gb["token_type_ids"],
                       pixel_values=gb["pixel_values"], sep_idx=gb["sep_idx"], return_dict=True)
    ml = out.logits[torch.arange(B, device=DEV), mi.cuda()][:, ids.cuda()]
    e_t = (trans.cpu() - trans_ref).abs().max().item()
    e_l = (ml.cpu() - ml_ref).abs().max().item()
    print(f"\nfp32 path patch {patch} B {B} conditioned={conditioned}: max|d trans| {e_t:.2e}  max|d logit| {e_l:.2e} (|logit| max {ml_ref.abs().max():.2f})")
    assert e_l < 1e-3, "BASELINE.json north_star: logits within 1e-3 of the fp32 reference path"
    assert e_t < 5e-3
    # ranked indices through the trainer surface (validation_step -> mart_rank): exact
    got = lit.validation_step({k: v for k, v in gb.items() if k in ("input_ids", "attention_mask", "token_type_ids", "pixel_values", "sep_idx", "label")}, 0)
    ref_ranks = np.asarray(O.ranks_count(ml_ref, batch["label"]))
    lab = ml_ref[torch.arange(B), batch["label"]]
    gap = (ml_ref - lab[:, None]).abs()
    gap[torch.arange(B), batch["label"]] = 1e9
    safe = (gap.min(1).values > 2 * e_l).numpy()
    print(f"   ranks hip {got['entity_ranks'].tolist()} oracle {ref_ranks.tolist()} (rows with a margin > 2x error: {int(safe.sum())}/{B})")
    assert np.array_equal(got["entity_ranks"][safe], ref_ranks[safe])
    assert np.abs(got["entity_ranks"] - ref_ranks).max() <= 2          # unsafe rows: the label sits within 2x error of a neighbour
    # training in this mode is refused loudly
    model.train()
    with pytest.raises(NotImplementedError):
        model(input_ids=gb["input_ids"], attention_mask=gb["attention_mask"], token_type_ids=gb["token_type_ids"],
              pixel_values=gb["pixel_values"], sep_idx=gb["sep_idx"], return_dict=True)
    model.set_precision("bf16")


def test_fp32_path_with_image_table():
    from mkg_analogy_amd import data_synth as D
    model, lit, cfg, vc, sd = _setup(32, 5, False)
    B = 3
    batch = D.make_batch(B, 64, seed=4)
    table = torch.randn(7, 3, 224, 224, generator=torch.Generator().manual_seed(2))
    index = torch.tensor([[0, 6], [3, -1], [-1, -1]], dtype=torch.int32)
    pix = torch.stack([torch.stack([table[i] if i >= 0 else torch.zeros(3, 224, 224) for i in row]) for row in index.tolist()])
    model.eval(); model.set_precision("fp32"); model.set_image_table(table)
    kw = dict(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], token_type_ids=batch["token_type_ids"],
              sep_idx=batch["sep_idx"], return_dict=True)
    _, t1 = model(image_index=index, **kw)
    _, t2 = model(pixel_values=pix, **kw)
    assert torch.equal(t1, t2)
    model.set_precision("bf16")


def test_fused_split_outputs_equal_the_split_pass():
    """mart_ln_fwd.out_split3 and mart_attn_fwd_f32.ctx_split3 (evaluation passes): the [hi | lo | hi] A operand written by the producer is
    bit-identical to mart_split_bf16x3 of the producer's f32 output."""
    from mkg_analogy_amd import ops
    g = torch.Generator().manual_seed(3)
    M, H = 1000, 768
    x = (2.0 * torch.randn(M, H, generator=g)).to(DEV)
    gam, bet = (1 + 0.1 * torch.randn(H, generator=g)).to(DEV), (0.1 * torch.randn(H, generator=g)).to(DEV)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    y, y3 = torch.empty(M, H, device=DEV), torch.empty(M, 3 * H, device=DEV, dtype=torch.bfloat16)
    ops.ln_fwd(x_f32=x, gamma=gam, beta=bet, eps=1e-5, M=M, H=H, mean=mean, rstd=rstd, out_f32=y)
    ops.ln_fwd(x_f32=x, gamma=gam, beta=bet, eps=1e-5, M=M, H=H, mean=mean, rstd=rstd, out_split3=y3)
    assert torch.equal(y3, ops.split_bf16x3(y, 0, terms=2))
    B, nh, Nv = 2, 12, 393
    qkv = torch.randn(B * Nv, 3 * H, generator=g).to(DEV)
    kw = dict(q=qkv[:, :H], k=qkv[:, H:2 * H], v=qkv[:, 2 * H:], B=B, nh=nh, D=64, Sq=Nv, Sk=Nv, scale=0.125, fast=True)
    c, c3 = torch.empty(B * Nv, H, device=DEV), torch.empty(B * Nv, 3 * H, device=DEV, dtype=torch.bfloat16)
    ops.attn_fwd_f32(ctx=c, **kw)
    ops.attn_fwd_f32(ctx=None, ctx_split3=c3, **kw)
    assert torch.equal(c3, ops.split_bf16x3(c, 0, terms=2))
    with pytest.raises(Exception):                       # the exact kernels do not write the operand form
        ops.attn_fwd_f32(ctx=None, ctx_split3=c3, **{**kw, "fast": False})


@pytest.mark.parametrize("act", ["none", "qgelu", "gelu"])
def test_gemm_nt_split3_output(act):
    """mart_gemm_nt_desc.c_split3: the epilogue writes the f32 result as the [hi | lo | hi] operand of the next split GEMM.  Without an activation it is
    bit-identical to mart_split_bf16x3 of the f32-output GEMM; with one, hi + lo reproduces the f32-output result to the activation's fast-form accuracy
    (rcp / exp / erf approximations of the 16-bit epilogue lanes: 1e-6 relative)."""
    from mkg_analogy_amd import ops
    g = torch.Generator().manual_seed(11)
    M, N, K = 1000, 768, 768
    a = ops.split_bf16x3(torch.randn(M, K, generator=g).to(DEV), 0, terms=2)
    w = ops.split_bf16x3((0.05 * torch.randn(N, K, generator=g)).to(DEV), 1, terms=2)
    bias = (0.1 * torch.randn(N, generator=g)).to(DEV)
    kind = dict(none=ops.ACT_NONE, qgelu=ops.ACT_QGELU, gelu=ops.ACT_GELU)[act]
    ref = torch.empty(M, N, device=DEV)
    ops.gemm_nt(a, w, ref, bias=bias, act=kind, tile_cfg=256)
    out = torch.full((M, 3 * N), float("nan"), device=DEV, dtype=torch.bfloat16)
    ops.gemm_nt(a, w, out, bias=bias, act=kind, tile_cfg=256, c_split3=True)
    assert bool(torch.isfinite(out.float()).all()) and torch.equal(out[:, :N], out[:, 2 * N:])
    if act == "none":
        assert torch.equal(out, ops.split_bf16x3(ref, 0, terms=2))
    rec = out[:, :N].float() + out[:, N:2 * N].float()
    err = float((rec - ref).abs().max()) / float(ref.abs().max())
    print(f"\nsplit3 epilogue ({act}): hi + lo vs the f32-output GEMM, max rel err {err:.2e}")
    assert err < 2e-5          # the two-term split itself carries 2^-16


@pytest.mark.parametrize("skip0", [False, True])
def test_attn_split_kernel_text_options(skip0):
    """attn_split_fwd_k with the text options (scale, adaptive reweight incl. the FLAVA row-0 variant, additive key mask), L = 64 and a ragged L = 40,
    against float64; the context written both as f32 and as the split operand."""
    from mkg_analogy_amd import ops
    g = torch.Generator().manual_seed(21)
    B, nh, H = 3, 12, 768
    for L in (64, 40):
        tq = torch.randn(B * L, 3 * H, generator=g)
        mask = torch.ones(B, L, dtype=torch.int64); mask[0, 33:] = 0; mask[2, 20:] = 0
        sep = torch.tensor([[3, 5, 17, 20, 25, 30], [2, 4, 9, 12, 14, 16], [1, 2, 11, 13, 15, 19]])
        w0, w1 = torch.tensor([0.31]), torch.tensor([0.77])
        dt = tq.to(DEV)
        c, c3 = torch.full((B * L, H), float("nan"), device=DEV), torch.empty(B * L, 3 * H, device=DEV, dtype=torch.bfloat16)
        ops.attn_fwd_f32(q=dt[:, :H], k=dt[:, H:2 * H], v=dt[:, 2 * H:], ctx=c, ctx_split3=c3, B=B, nh=nh, D=64, Sq=L, Sk=L, scale=0.125,
                         attn_mask=mask.to(DEV), sep=sep.to(DEV)[:, 2:], sep_stride=6, w0=w0.to(DEV), w1=w1.to(DEV), rw_skip_row0=skip0, fast=True)
        ref = _ref_attn(tq[:, :H].reshape(B, L, H), tq[:, H:2 * H].reshape(B, L, H), tq[:, 2 * H:].reshape(B, L, H), nh, 0.125,
                        mask=mask, sep=sep[:, 2], w0=w0, w1=w1)
        if skip0:                                             # FLAVA: query row 0 keeps factor 1 (flava_oracle / modeling_flava.py reweight variant)
            ref0 = _ref_attn(tq[:, :H].reshape(B, L, H), tq[:, H:2 * H].reshape(B, L, H), tq[:, 2 * H:].reshape(B, L, H), nh, 0.125, mask=mask)
            ref[:, 0] = ref0[:, 0]
        e = (c.cpu().double().view(B, L, H) - ref).abs().max().item()
        print(f"\nsplit attention, text options, L = {L}, skip_row0 = {skip0}: max|err| {e:.2e} (|out| max {ref.abs().max():.2f})")
        assert e < 2.5e-5 * max(1.0, float(ref.abs().max()))
        assert torch.equal(c3, ops.split_bf16x3(c, 0, terms=2))
