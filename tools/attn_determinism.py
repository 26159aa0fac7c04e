"""Run attn_bwd twice on identical inputs and compare every output bitwise (vision shapes: two-pass kernels at Sq=99, fused at 393)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mkg_analogy_amd import ops
ops.require_gpu()
DEV, BF = "cuda", torch.bfloat16
for (B, S, Lp) in [(16, 99, 64), (16, 99, 0), (16, 393, 64), (16, 393, 0), (16, 458, 0)]:
    nh, H = 12, 768
    torch.manual_seed(0)
    qkv = torch.randn(B * S, 3 * H, device=DEV).to(BF)
    tq = torch.randn(B * 64, 3 * H, device=DEV).to(BF)
    dctx = torch.randn(B * S, H, device=DEV).to(BF)
    ctx = torch.empty(B * S, H, device=DEV, dtype=BF)
    lse = torch.empty(B, nh, S, device=DEV)
    kw = dict(q=qkv[:, :H], k=qkv[:, H:2*H], v=qkv[:, 2*H:], ctx=ctx, lse=lse, B=B, nh=nh, Sq=S, Sk=S, scale=0.125,
              pk=tq[:, H:2*H] if Lp else None, pv=tq[:, 2*H:] if Lp else None, Lp=Lp)
    ops.attn_fwd(**kw)
    outs = []
    for r in range(3):
        dqkv = torch.zeros(B * S, 3 * H, device=DEV, dtype=BF)
        dt = torch.zeros(B * 64, 3 * H, device=DEV, dtype=BF)
        delta = torch.empty(B, nh, S, device=DEV)
        ops.attn_bwd(dctx=dctx, delta=delta, dq=dqkv[:, :H], dk=dqkv[:, H:2*H], dv=dqkv[:, 2*H:],
                     dpk=dt[:, H:2*H] if Lp else None, dpv=dt[:, 2*H:] if Lp else None, **kw)
        torch.cuda.synchronize()
        outs.append((dqkv.clone(), dt.clone(), delta.clone()))
    for r in (1, 2):
        d = [int((outs[0][i] != outs[r][i]).sum()) for i in range(3)]
        dq = int((outs[0][0][:, :H] != outs[r][0][:, :H]).sum()); dk = int((outs[0][0][:, H:2*H] != outs[r][0][:, H:2*H]).sum()); dv = int((outs[0][0][:, 2*H:] != outs[r][0][:, 2*H:]).sum())
        print(f"B={B} S={S} Lp={Lp} run {r}: differing elements dq {dq} dk {dk} dv {dv} prefix {d[1]} delta {d[2]}   nan {int(torch.isnan(outs[r][0].float()).sum())}")
