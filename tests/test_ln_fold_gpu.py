"""LayerNorm folded into the product that consumes it (mart_gemm_nt_desc.row_stats / ln_*, mart_ln_fold_prep, mart_ln_stats_finalize):
LN(x) W^T + b = rstd (x (gamma o W)^T - mean s) + b'.  Each piece against plain PyTorch fp32, through the C ABI.  Reference lines:
nn.LayerNorm -> nn.Linear at modeling_unimo.py:509 -> :223-225 (layer_norm1 -> q/k/v) and :518 -> :284-286 (layer_norm2 -> fc1)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"
BF, F32 = torch.bfloat16, torch.float32


@pytest.fixture(scope="module")
def ops():
    from mkg_analogy_amd import ops as o
    o.require_gpu()
    return o


def rnd(*shape, scale=1.0, seed=0, dtype=F32):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(dtype)


@pytest.mark.parametrize("M,cfg", [(33 * 256, 0), (2500, 256), (1000, 128), (33 * 256 + 77, 0)])
def test_fold_producer_consumer(ops, M, cfg):
    H, eps = 768, 1e-5
    ctx, wo, bo = rnd(M, H, seed=1, dtype=BF), rnd(H, H, seed=2, scale=0.03, dtype=BF), rnd(H, seed=3, scale=0.1)
    xv = rnd(M, H, seed=4, scale=1.5) + 0.2                                   # residual stream with a row mean that is not zero
    gamma, beta = 1 + 0.2 * rnd(H, seed=5), 0.1 * rnd(H, seed=6)
    # ---- producer: out-proj + f32 residual, bf16 copy, per-row partial sums
    x1, x1b = torch.empty(M, H, device=DEV), torch.empty(M, H, device=DEV, dtype=BF)
    part = torch.full((M, H // 64, 2), float("nan"), device=DEV)
    ops.gemm_nt(ctx, wo, x1, bias=bo, res_f32=xv, C2=x1b, row_stats=part, tile_cfg=cfg)
    ref = ctx.float() @ wo.float().t() + bo + xv
    assert float((x1 - ref).abs().max()) < 2e-3
    assert torch.equal(x1b, x1.to(BF))
    x1c = torch.empty_like(x1)
    ops.gemm_nt(ctx, wo, x1c, bias=bo, res_f32=xv, tile_cfg=cfg)
    assert torch.equal(x1, x1c), "the statistics option must not change the product"
    cols = x1.view(M, H // 64, 64)
    assert float((part[..., 0] - cols.sum(-1)).abs().max()) < 2e-4 and float((part[..., 1] - (cols * cols).sum(-1)).abs().max()) < 2e-3
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    ops.ln_stats_finalize(part, M, H, eps, mean, rstd)
    mu, var = x1.double().mean(1), x1.double().var(1, unbiased=False)
    assert float((mean - mu).abs().max()) < 1e-5
    assert float((rstd / torch.rsqrt(var + eps) - 1).abs().max()) < 2e-5
    # ---- operands of the fold
    for N, act in ((2304, ops.ACT_NONE), (3072, ops.ACT_QGELU)):
        W, b = rnd(N, H, seed=7, scale=0.03), rnd(N, seed=8, scale=0.1)
        Wf, s, bfold = torch.empty(N, H, device=DEV, dtype=BF), torch.empty(N, device=DEV), torch.empty(N, device=DEV)
        ops.ln_fold_prep(W, b, gamma, beta, Wf, s, bfold)
        assert torch.equal(Wf, (W * gamma).to(BF))
        assert float((s - Wf.float().sum(1)).abs().max()) < 1e-4 and float((bfold - (b + W @ beta)).abs().max()) < 1e-4
        # ---- consumer: the bf16 copy of x times the folded weight, statistics applied in the epilogue
        y = torch.empty(M, N, device=DEV, dtype=BF)
        ops.gemm_nt(x1b, Wf, y, bias=bfold, act=act, ln_mean=mean, ln_rstd=rstd, ln_colsum=s, tile_cfg=cfg)
        z = torch.nn.functional.layer_norm(x1, (H,), gamma, beta, eps) @ W.t() + b
        want = z * torch.sigmoid(1.702 * z) if act == ops.ACT_QGELU else z
        # same operands, fp32 arithmetic: what the kernel computes up to accumulation order and the bf16 rounding of the result
        zk = ((x1b.float() @ Wf.float().t()) - mean[:, None] * s[None, :]) * rstd[:, None] + bfold
        wk = zk * torch.sigmoid(1.702 * zk) if act == ops.ACT_QGELU else zk
        err_k = (y.float() - wk).abs()
        assert float((err_k - 8e-3 * wk.abs()).max()) < 4e-3, float(err_k.max())
        # against LayerNorm -> Linear in fp32: the bf16 rounding of x and of gamma o W (the unfused bf16 path rounds LN(x) and W instead)
        hb = torch.nn.functional.layer_norm(x1, (H,), gamma, beta, eps).to(BF)
        zu = hb.float() @ W.to(BF).float().t() + b
        wu = zu * torch.sigmoid(1.702 * zu) if act == ops.ACT_QGELU else zu
        e_fold, e_unf = float((y.float() - want).pow(2).mean().sqrt()), float((wu - want).pow(2).mean().sqrt())
        print(f"M {M} N {N}: rms error vs fp32 LayerNorm -> Linear: folded {e_fold:.3e}, unfused bf16 operands {e_unf:.3e} (result rms {float(want.pow(2).mean().sqrt()):.3f})")
        assert e_fold < 1.5 * e_unf + 4e-3
