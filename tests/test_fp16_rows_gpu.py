"""Round-4 kernel additions against plain PyTorch fp32 references of the same ops (through the C ABI):
  * fp16 operands in mart_gemm_nt (in_f16), fp16 C with its bf16 copy C2 (c_f16), every fast epilogue the text stream uses + the general one,
  * fp16 twins of the LayerNorm / text-embedding / attention-context / fusion outputs, the AdamW fp16 shadow,
  * the row-subset helpers of the last text layer (x_rows gather in mart_ln_fwd, mart_gather_rows_first_f32, mart_scatter_rows)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"
BF, F32, HF = torch.bfloat16, torch.float32, torch.float16


@pytest.fixture(scope="module")
def ops():
    from mkg_analogy_amd import ops as o
    o.require_gpu()
    return o


def rnd(*shape, scale=1.0, seed=0, dtype=F32):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(dtype)


def close(got, ref, atol, rtol, what=""):
    got, ref = got.float(), ref.float()
    err = (got - ref).abs()
    bad = err > atol + rtol * ref.abs()
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} bad, max err {float(err.max()):.4g}, ref max {float(ref.abs().max()):.4g}"


def gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x * 0.7071067811865476))


def dgelu(x):
    return 0.5 * (1.0 + torch.erf(x * 0.7071067811865476)) + x * torch.exp(-0.5 * x * x) * 0.3989422804014327


@pytest.mark.parametrize("M,N,K,cfg", [(1000, 768, 768, 128), (2048, 768, 768, 256), (16384, 2304, 768, 0), (1280, 3072, 768, 0), (300, 200, 128, 0), (160, 768, 3072, 0)])
def test_gemm_nt_fp16_operands(ops, M, N, K, cfg):
    """fp16 operands: 8x finer rounding than bf16 at the same rate -- the f32 result is held to 4e-4 of the exact product of the fp16 values
    and is closer to the f32 product of the UNROUNDED operands than the bf16 kernel is."""
    A32, B32 = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    bias = rnd(N, seed=3, scale=0.1)
    A, B = A32.to(HF), B32.to(HF)
    ref = A.float() @ B.float().t() + bias
    out = torch.empty(M, N, device=DEV, dtype=F32)
    ops.gemm_nt(A, B, out, bias=bias, tile_cfg=cfg)
    close(out, ref, 4e-4, 4e-4, "fp16 operands, f32 out")
    outb = torch.empty(M, N, device=DEV, dtype=BF)
    ops.gemm_nt(A, B, outb, bias=bias, tile_cfg=cfg)
    close(outb, ref, 2e-2, 1e-2, "fp16 operands, bf16 out")
    exact = A32 @ B32.t() + bias
    outbf = torch.empty(M, N, device=DEV, dtype=F32)
    ops.gemm_nt(A32.to(BF), B32.to(BF), outbf, bias=bias, tile_cfg=cfg)
    e_h, e_b = float((out - exact).pow(2).mean().sqrt()), float((outbf - exact).pow(2).mean().sqrt())
    print(f"\n[{M}x{N}x{K}] rms error vs the f32 product of the unrounded operands: fp16 operands {e_h:.3e}, bf16 operands {e_b:.3e}")
    assert e_h < 0.25 * e_b


@pytest.mark.parametrize("M,cfg", [(16384, 0), (1280, 0), (300, 0), (2048, 2561)])
@pytest.mark.parametrize("dual", [False, True])
def test_gemm_nt_fp16_gelu_epilogue_three_outputs(ops, M, cfg, dual):
    """intermediate.dense (+ fusion_dense as a second K segment): C = fp16 GELU output, C2 = its bf16 copy, preact = act'(z) in bf16; and
    the no_grad form (fp16 C only).  (2561 = the general epilogue.)"""
    N, K = 3072, 768
    A, B = rnd(M, K, seed=1).to(HF), rnd(N, K, seed=2, scale=0.04).to(HF)
    A2, B2 = (rnd(M, K, seed=5).to(HF), rnd(N, K, seed=6, scale=0.04).to(HF)) if dual else (None, None)
    b1, b2 = rnd(N, seed=3, scale=0.1), (rnd(N, seed=4, scale=0.1) if dual else None)
    z = A.float() @ B.float().t() + b1
    if dual:
        z = z + A2.float() @ B2.float().t() + b2
    c, c2, pg = torch.empty(M, N, device=DEV, dtype=HF), torch.empty(M, N, device=DEV, dtype=BF), torch.empty(M, N, device=DEV, dtype=BF)
    ops.gemm_nt(A, B, c, A2=A2, B2=B2, bias=b1, bias2=b2, act=ops.ACT_GELU, preact=pg, preact_grad=True, C2=c2, tile_cfg=cfg)
    close(c, gelu(z), 2e-3, 1.5e-3, "fp16 GELU output")
    close(c2, gelu(z), 2e-2, 1e-2, "bf16 copy")
    close(pg, dgelu(z), 2e-2, 1e-2, "act'(z)")
    d2 = (c2.float() - c.float()).abs()
    assert bool((d2 <= 2.0 ** -8 * c.float().abs() + 1e-6).all()), "C2 is the same value as C, rounded to bf16 instead of fp16"
    c_ng = torch.empty(M, N, device=DEV, dtype=HF)
    ops.gemm_nt(A, B, c_ng, A2=A2, B2=B2, bias=b1, bias2=b2, act=ops.ACT_GELU, tile_cfg=cfg)
    close(c_ng, gelu(z), 2e-3, 1.5e-3, "fp16 GELU output, no_grad form")


def test_gemm_nt_fp16_row_gather_into_f32(ops):
    """attention.output.dense of the last text layer: the A rows are gathered (a_rows) from the [B * L, H] fp16 context."""
    Mt, H, R = 4096, 768, 320
    A, B = rnd(Mt, H, seed=1).to(HF), rnd(H, H, seed=2, scale=0.04).to(HF)
    rows = torch.randperm(Mt, generator=torch.Generator().manual_seed(0))[:R].to(torch.int32).to(DEV)
    bias = rnd(H, seed=3, scale=0.1)
    out = torch.empty(R, H, device=DEV, dtype=F32)
    ops.gemm_nt(A, B, out, a_rows=rows, bias=bias)
    close(out, A[rows.long()].float() @ B.float().t() + bias, 4e-4, 4e-4, "gathered rows")


def test_gemm_nt_mixed_types_are_rejected(ops):
    from mkg_analogy_amd._lib import MartError
    A, B = rnd(256, 128).to(HF), rnd(256, 128).to(HF)
    with pytest.raises(AssertionError):
        ops.gemm_nt(A, B.to(BF), torch.empty(256, 256, device=DEV, dtype=F32))
    with pytest.raises(AssertionError):
        ops.gemm_nt(A.to(BF), B.to(BF), torch.empty(256, 256, device=DEV, dtype=HF))
    with pytest.raises(MartError):                                       # fp16 operands are a forward-pass option
        ops.gemm_nt(A, B, torch.empty(256, 256, device=DEV, dtype=BF), mulz=torch.zeros(256, 256, device=DEV, dtype=BF), mul_act=ops.ACT_STORED)


@pytest.mark.parametrize("gather", [False, True])
def test_ln_fwd_fp16_twin_and_row_gather(ops, gather):
    M0, H, M = 4096, 768, 1280
    x = rnd(M0, H, seed=1)
    y = rnd(M if gather else M0, H, seed=2, scale=0.3)
    rows = torch.randint(0, M0, (M,), generator=torch.Generator().manual_seed(1)).to(torch.int32).to(DEV) if gather else None
    Mx = M if gather else M0
    gamma, beta = 1 + 0.1 * rnd(H, seed=3), 0.1 * rnd(H, seed=4)
    o32, ob, oh = torch.empty(Mx, H, device=DEV), torch.empty(Mx, H, device=DEV, dtype=BF), torch.empty(Mx, H, device=DEV, dtype=HF)
    s = torch.empty(Mx, H, device=DEV)
    mean, rstd = torch.empty(Mx, device=DEV), torch.empty(Mx, device=DEV)
    ops.ln_fwd(x_f32=x, x_rows=rows, y_f32=y, gamma=gamma, beta=beta, eps=1e-12, M=Mx, H=H, mean=mean, rstd=rstd, s_out=s, out_f32=o32, out_bf16=ob, out_f16=oh)
    xs = (x[rows.long()] if gather else x) + y
    ref = torch.nn.functional.layer_norm(xs, (H,), gamma, beta, 1e-12)
    close(s, xs, 1e-6, 1e-6, "pre-LN sum")
    close(o32, ref, 2e-5, 2e-5, "f32 out")
    assert torch.equal(ob, o32.to(BF)) and torch.equal(oh, o32.to(HF)), "the 16-bit twins are the f32 output rounded once"
    # dropout indexes the COMPACT row: same masks as a dense call on the same compact rows
    o2 = torch.empty(Mx, H, device=DEV)
    ops.ln_fwd(x_f32=x, x_rows=rows, y_f32=y, gamma=gamma, beta=beta, eps=1e-12, M=Mx, H=H, mean=mean, rstd=rstd, out_f32=o2, p_drop=0.1, seed=77)
    xc = (x[rows.long()] if gather else x).contiguous()
    o3 = torch.empty(Mx, H, device=DEV)
    ops.ln_fwd(x_f32=xc, y_f32=y, gamma=gamma, beta=beta, eps=1e-12, M=Mx, H=H, mean=mean, rstd=rstd, out_f32=o3, p_drop=0.1, seed=77)
    assert torch.equal(o2, o3)


def test_text_embed_fp16_twin(ops):
    B, L, H, V = 4, 64, 768, 1000
    ids = torch.randint(0, V, (B, L), device=DEV)
    tt = torch.randint(0, 2, (B, L), device=DEV)
    word, pos, typ = rnd(V, H, seed=1, scale=0.02), rnd(512, H, seed=2, scale=0.02), rnd(2, H, seed=3, scale=0.02)
    gamma, beta = 1 + 0.1 * rnd(H, seed=4), 0.1 * rnd(H, seed=5)
    M = B * L
    s, mean, rstd = torch.empty(M, H, device=DEV), torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    o32, ob, oh = torch.empty(M, H, device=DEV), torch.empty(M, H, device=DEV, dtype=BF), torch.empty(M, H, device=DEV, dtype=HF)
    ops.text_embed_fwd(ids=ids, tt=tt, word=word, pos=pos, type_=typ, gamma=gamma, beta=beta, eps=1e-12, p_drop=0.0, seed=1, B=B, Lq=L, H=H,
                       s_out=s, mean=mean, rstd=rstd, out_f32=o32, out_bf16=ob, out_f16=oh)
    ref = torch.nn.functional.layer_norm(word[ids] + typ[tt] + pos[:L][None], (H,), gamma, beta, 1e-12).view(M, H)
    close(o32, ref, 2e-5, 2e-5, "text embeddings")
    assert torch.equal(oh, o32.to(HF)) and torch.equal(ob, o32.to(BF))


def test_attention_and_fusion_fp16_twins(ops):
    """ctx_f16 / out_f16: the same values as the bf16 outputs, rounded once from the f32 accumulators to fp16 (closer to an f32 reference)."""
    B, nh, L, H = 3, 12, 64, 768
    qkv = rnd(B * L, 3 * H, seed=1, scale=0.5).to(BF)
    ctx, ctxh = torch.empty(B * L, H, device=DEV, dtype=BF), torch.empty(B * L, H, device=DEV, dtype=HF)
    lse = torch.empty(B, nh, L, device=DEV)
    am = torch.ones(B, L, dtype=torch.int64, device=DEV)
    am[1, 50:] = 0
    ops.attn_fwd(q=qkv[:, :H], k=qkv[:, H:2 * H], v=qkv[:, 2 * H:], ctx=ctx, lse=lse, B=B, nh=nh, Sq=L, Sk=L, scale=0.125, attn_mask=am, ctx_f16=ctxh)
    q, k, v = (qkv[:, i * H:(i + 1) * H].float().view(B, L, nh, 64).transpose(1, 2) for i in range(3))
    sc = q @ k.transpose(-1, -2) * 0.125 + ((1 - am.float()) * -10000.0)[:, None, None, :]
    ref = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(B * L, H)
    close(ctx, ref, 2e-2, 2e-2, "bf16 context")
    close(ctxh, ref, 1e-2, 1e-2, "fp16 context")
    assert float((ctxh.float() - ctx.float()).abs().max()) <= float(ctx.float().abs().max()) * 2 ** -8
    assert float((ctxh.float() - ref).pow(2).mean()) < float((ctx.float() - ref).pow(2).mean())
    Nv = 393
    if not ops.fusion_supported(L, Nv, H):
        pytest.skip("fused fusion kernel not available for this shape")
    hid, vis = rnd(B * L, H, seed=2, scale=0.05).to(BF), rnd(B * Nv, H, seed=3).to(BF)
    out, outh = torch.empty(B * L, H, device=DEV, dtype=BF), torch.empty(B * L, H, device=DEV, dtype=HF)
    probs = torch.empty(B * L, 448, device=DEV, dtype=BF)
    ops.fusion_fwd(hid, vis, out, probs, B, L, Nv, H, out_f16=outh)
    hv, vv = hid.float().view(B, L, H), vis.float().view(B, Nv, H)
    reff = (torch.softmax(hv @ vv.transpose(1, 2), -1) @ vv).reshape(B * L, H)
    close(out, reff, 3e-2, 2e-2, "fusion bf16")
    close(outh, reff, 3e-2, 2e-2, "fusion fp16")
    assert float((outh.float() - out.float()).abs().max()) <= float(out.float().abs().max()) * 2 ** -8


def test_adamw_refreshes_the_fp16_shadow_of_flagged_chunks(ops):
    n = 3 * 65536
    g = torch.Generator().manual_seed(0)
    master = (torch.randn(n, generator=g) * 0.02).to(DEV)
    grad = (torch.randn(n, generator=g) * 1e-3).to(DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    sh, sh16 = torch.zeros(n, device=DEV, dtype=BF), torch.zeros(n, device=DEV, dtype=HF)
    chunks = torch.tensor([[0, 65536, 1], [65536, 65536, 3], [131072, 65536, 2]], dtype=torch.int32, device=DEV)   # decay / decay + f16 / f16 only
    w0 = master.clone()
    ops.adamw(master=master, grad=grad, m=m, v=v, shadow=sh, shadow_f16=sh16, chunks=chunks, n_chunks=3, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8,
              weight_decay=0.01, bc1=0.1, bc2=0.001)
    ref = torch.optim.AdamW
    p = torch.nn.Parameter(w0.clone())
    p.grad = grad.clone()
    o1 = ref([{"params": [p]}], lr=1e-3, eps=1e-8, weight_decay=0.01)
    o1.step()
    close(master[:131072], p.data[:131072], 1e-7, 1e-5, "decayed chunks")
    assert torch.equal(sh, master.to(BF))
    assert torch.equal(sh16[65536:], master[65536:].to(HF)) and float(sh16[:65536].abs().max()) == 0.0
    p2 = torch.nn.Parameter(w0.clone())
    p2.grad = grad.clone()
    torch.optim.AdamW([{"params": [p2]}], lr=1e-3, eps=1e-8, weight_decay=0.0).step()
    close(master[131072:], p2.data[131072:], 1e-7, 1e-5, "no-decay chunk")


def test_row_subset_helpers(ops):
    B, nr, L, H = 8, 5, 64, 768
    g = torch.Generator().manual_seed(3)
    pos = torch.randint(0, L, (B, nr), generator=g)
    pos[0, 3] = pos[0, 1]                                   # a row requested twice
    pos[5, 4] = pos[5, 0]
    rows = (torch.arange(B)[:, None] * L + pos).to(torch.int32).to(DEV).contiguous()
    R = rows.view(-1)
    dense = rnd(B * L, H, seed=1)
    # gradient gather: a repeated row is taken once
    got = torch.empty(B * nr, H, device=DEV)
    ops.gather_rows_first_f32(dense, R, nr, got)
    ref = dense[R.long()].clone()
    Rl = R.cpu().tolist()
    dups = [r for r in range(B * nr) if Rl[r] in Rl[(r // nr) * nr:r]]        # an earlier slot of the same example names the same row
    assert 0 * nr + 3 in dups and 5 * nr + 4 in dups
    ref[dups] = 0
    assert torch.equal(got, ref)
    # forward scatter: first slot wins, the rest of the tensor keeps its zeros
    vals = rnd(B * nr, H, seed=2)
    for dt in (F32, BF):
        out = torch.zeros(B * L, H, device=DEV, dtype=dt)
        ops.scatter_rows(vals, R, nr, out)
        exp = torch.zeros(B * L, H, device=DEV)
        for r in reversed(range(B * nr)):                  # reversed: the first slot is written last = wins
            exp[int(R[r])] = vals[r]
        assert torch.equal(out, exp.to(dt))
    # gradient scatter: += in slot order (a repeated row receives both), f32 and bf16 (summed in f32, rounded once per slot)
    base = rnd(B * L, H, seed=4)
    out = base.clone()
    ops.scatter_rows(vals, R, nr, out, accumulate=True)
    exp = base.clone()
    for r in range(B * nr):
        exp[int(R[r])] += vals[r]
    assert torch.equal(out, exp)
    outb = base.to(BF)
    ops.scatter_rows(vals, R, nr, outb, accumulate=True)
    expb = base.to(BF)
    for r in range(B * nr):
        expb[int(R[r])] = (expb[int(R[r])].float() + vals[r]).to(BF)
    assert torch.equal(outb, expb)


@pytest.mark.parametrize("M", [4096 + 3, 1000])
def test_ln_bwd_second_residual_operand(ops, M):
    """mart_ln_bwd.add2_f32 (the fusion op's d(visual) side buffer entering the vision LayerNorm-1 backward): the same as adding it to add_f32
    beforehand, for the straight-line vision-stream kernel (M >= 4096) and the general one."""
    H = 768
    dy, x = rnd(M, H, seed=1).to(BF), rnd(M, H, seed=2)
    a1, a2 = rnd(M, H, seed=3), rnd(M, H, seed=4, scale=0.1)
    gamma = 1 + 0.1 * rnd(H, seed=5)
    mean, rstd = x.mean(1), (x.var(1, unbiased=False) + 1e-5).rsqrt()
    outs = []
    for add, add2 in ((a1, a2), (a1 + a2, None)):
        ds, dsb = torch.empty(M, H, device=DEV), torch.empty(M, H, device=DEV, dtype=BF)
        dg, db = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
        ops.ln_bwd(dy_bf16=dy, s=x, mean=mean, rstd=rstd, gamma=gamma, M=M, H=H, add_f32=add, add2_f32=add2, ds_f32=ds, ds_bf16=dsb, bf16_total=True,
                   dgamma=dg, dbeta=db)
        outs.append((ds, dsb, dg, db))
    close(outs[0][0], outs[1][0], 2e-6, 2e-6, "ds with the second residual operand")
    assert torch.equal(outs[0][2], outs[1][2]) and torch.equal(outs[0][3], outs[1][3])
    assert float((outs[0][1].float() - outs[1][1].float()).abs().max()) <= 2.0 ** -7 * float(outs[1][0].abs().max())


@pytest.mark.parametrize("M", [4096 + 3, 1000])
def test_ln_bwd_deferred_reduction_is_the_same_sum(ops, M):
    """mart_ln_bwd.defer_reduce + mart_ln_dgb_reduce (the dgamma / dbeta reduction issued by the caller, on another stream): bit-identical to the
    in-order form -- same partials, same kernel, same order."""
    H = 768
    dy, x, a1 = rnd(M, H, seed=1).to(BF), rnd(M, H, seed=2), rnd(M, H, seed=3)
    gamma = 1 + 0.1 * rnd(H, seed=5)
    mean, rstd = x.mean(1), (x.var(1, unbiased=False) + 1e-5).rsqrt()
    outs = []
    side = torch.cuda.Stream()
    for defer in (False, True):
        ds, dsb = torch.empty(M, H, device=DEV), torch.empty(M, H, device=DEV, dtype=BF)
        dg, db = torch.full((H,), 0.5, device=DEV), torch.full((H,), -0.25, device=DEV)      # the reduction ADDS into the gradient buffers
        r = ops.ln_bwd(dy_bf16=dy, s=x, mean=mean, rstd=rstd, gamma=gamma, M=M, H=H, add_f32=a1, ds_f32=ds, ds_bf16=dsb, bf16_total=True,
                       dgamma=dg, dbeta=db, defer_reduce=defer)
        if defer:
            ws, n = r
            assert 0 < n <= 768
            ev = torch.cuda.Event(); ev.record()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                ops.ln_dgb_reduce(ws, n, H, dg, db)
            torch.cuda.current_stream().wait_stream(side)
        else:
            assert r is None
        outs.append((ds, dg, db))
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    assert float((outs[0][1] - 0.5).abs().max()) > 0
