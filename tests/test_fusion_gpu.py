"""BertFusion as one kernel per direction (mart_fusion_fwd / mart_fusion_bwd) against a plain fp32 torch statement of
modeling_unimo.py:405-411 (scores = hidden @ visual^T, softmax, probs @ visual) and its autograd."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(q, v):
    s = torch.einsum("bqd,bkd->bqk", q, v)
    p = torch.softmax(s, dim=-1)
    return torch.einsum("bqk,bkd->bqd", p, v), p


def _rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-30))


@pytest.mark.parametrize("B,Lq,Nv", [(3, 64, 393), (2, 96, 393), (2, 32, 99), (2, 64, 448), (1, 64, 197), (2, 64, 33),
                                     # lengths real MARS batches have (the reference pads to the longest example of the batch, data_module.py:113-119):
                                     # the last 32-row query block is partial
                                     (3, 57, 393), (2, 101, 393), (3, 37, 99), (2, 5, 393), (1, 33, 197)])
def test_fusion_kernels_match_fp32_reference(B, Lq, Nv):
    from mkg_analogy_amd import ops
    dev, H = torch.device("cuda:0"), 768
    assert ops.fusion_supported(Lq, Nv, H)
    g = torch.Generator(device="cpu").manual_seed(1000 * Lq + Nv)
    q = (0.5 * torch.randn(B * Lq, H, generator=g)).to(dev, torch.bfloat16)
    v = (0.5 * torch.randn(B * Nv, H, generator=g)).to(dev, torch.bfloat16)
    dout = (torch.randn(B * Lq, H, generator=g)).to(dev, torch.bfloat16)
    base = torch.randn(B * Nv, H, generator=g).to(dev)
    Nvp = ((Nv + 63) // 64) * 64
    out = torch.empty(B * Lq, H, device=dev, dtype=torch.bfloat16)
    probs = torch.full((B * Lq, Nvp), 7.0, device=dev, dtype=torch.bfloat16)
    ops.fusion_fwd(q, v, out, probs, B, Lq, Nv, H)

    qf = q.float().view(B, Lq, H).requires_grad_(True)
    vf = v.float().view(B, Nv, H).requires_grad_(True)
    o_ref, p_ref = _ref(qf, vf)
    assert _rel(out.view(B, Lq, H), o_ref) < 6e-3                       # bf16 rounding of P and of the output
    assert float((probs.view(B, Lq, Nvp)[:, :, :Nv].float() - p_ref).abs().max()) < 4e-3
    assert float(probs.view(B, Lq, Nvp)[:, :, Nv:].float().abs().max()) == 0.0 if Nvp > Nv else True
    assert float((probs.float().sum(-1) - 1).abs().max()) < 2e-2

    o_ref.backward(dout.float().view(B, Lq, H))
    dq = torch.empty(B * Lq, H, device=dev, dtype=torch.bfloat16)
    dv = base.clone()
    dvb = torch.zeros(B * Nv, H, device=dev, dtype=torch.bfloat16)
    ops.fusion_bwd(q, v, dout, probs, dq, dv, dvb, B, Lq, Nv, H)
    assert _rel(dq.view(B, Lq, H), qf.grad) < 1.5e-2
    assert _rel(dv.view(B, Nv, H) - base.view(B, Nv, H), vf.grad) < 1.5e-2
    assert torch.equal(dvb, dv.to(torch.bfloat16))
    # in place, fixed order: a second run from the same state gives the same bits
    dq2, dv2 = torch.empty_like(dq), base.clone()
    ops.fusion_bwd(q, v, dout, probs, dq2, dv2, None, B, Lq, Nv, H)
    assert torch.equal(dq, dq2) and torch.equal(dv, dv2)
    # round 6: the vision-stream gradient carried in bf16 -- the read-modify-write goes to the bf16 tensor alone (dv_f32 = None)
    dq3 = torch.empty_like(dq)
    dvb3 = base.to(torch.bfloat16)
    ops.fusion_bwd(q, v, dout, probs, dq3, None, dvb3, B, Lq, Nv, H)
    assert torch.equal(dq3, dq)
    want = (base.to(torch.bfloat16).float() + (dv - base)).to(torch.bfloat16)        # bf16(old bf16 value + the f32 update), one rounding per 64-query block
    diff = (dvb3.float() - want.float()).abs()
    upd = (dv - base).abs()
    tol = 2.0 ** -7 * (base.abs() + upd) + (2.0 ** -8 * float(upd.max()) if Lq > 64 else 0.0) + 2e-3   # (64-query blocks are sequential launches: Lq > 64 rounds twice, at the magnitude of the intermediate sum)
    assert bool((diff <= tol).all()), float((diff - tol).max())


def test_fusion_shape_gate():
    from mkg_analogy_amd import ops
    assert ops.fusion_supported(64, 448, 768) and ops.fusion_supported(96, 393, 768)
    assert not ops.fusion_supported(64, 457, 768)       # dS + P images + the Q / dO chunks of the backward pass exceed 160 KB of LDS
    assert not ops.fusion_supported(64, 393, 1024)
    assert ops.fusion_supported(48, 393, 768) and ops.fusion_supported(57, 393, 768) and ops.fusion_supported(101, 99, 768)   # any Lq (round 6)
