import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mkg_analogy_amd import ops
ops.require_gpu()
DEV = "cuda"; BF = torch.bfloat16
B, nh, H, Nv, L = 256, 12, 768, 393, 64
torch.manual_seed(0)
bad_total = 0
junk = []
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    # churn the allocator with NaN-filled garbage so that uninitialised reads show up
    g = torch.full((64 << 20,), float("nan"), device=DEV, dtype=torch.float32); del g
    qkv = (torch.randn(B * Nv, 3 * H, device=DEV) * 0.5).to(BF)
    pre = (torch.randn(B * L, 3 * H, device=DEV) * 0.5).to(BF)
    ctx = torch.empty(B * Nv, H, device=DEV, dtype=BF); lse = torch.empty(B, nh, Nv, device=DEV)
    kw = dict(q=qkv[:, :H], k=qkv[:, H:2 * H], v=qkv[:, 2 * H:], ctx=ctx, lse=lse, B=B, nh=nh, Sq=Nv, Sk=Nv, scale=0.125,
              pk=pre[:, H:2 * H], pv=pre[:, 2 * H:], Lp=L)
    ops.attn_fwd(**kw)
    dctx = (torch.randn(B * Nv, H, device=DEV) * 0.1).to(BF)
    dqkv = torch.empty(B * Nv, 3 * H, device=DEV, dtype=BF); dpre = torch.empty(B * L, 3 * H, device=DEV, dtype=BF)
    delta = torch.empty(B, nh, Nv, device=DEV)
    ops.attn_bwd(dctx=dctx, delta=delta, dq=dqkv[:, :H], dk=dqkv[:, H:2 * H], dv=dqkv[:, 2 * H:], dpk=dpre[:, H:2 * H], dpv=dpre[:, 2 * H:], **kw)
    torch.cuda.synchronize()
    nb = [(~torch.isfinite(x.float())).sum().item() for x in (ctx, lse, delta, dqkv[:, :H], dqkv[:, H:2 * H], dqkv[:, 2 * H:], dpre[:, H:])]
    big = dqkv.float().abs().max().item()
    if any(nb) or big > 1e4:
        bad_total += 1
        print(f"iter {it}: nonfinite ctx {nb[0]} lse {nb[1]} delta {nb[2]} dq {nb[3]} dk {nb[4]} dv {nb[5]} dpre(k,v) {nb[6]} absmax {big:.3e}")
        if nb[3]:
            rows = (~torch.isfinite(dqkv[:, :H].float())).any(1).nonzero().flatten()
            cols = (~torch.isfinite(dqkv[:, :H].float())).any(0).nonzero().flatten()
            print("   dq bad rows", rows[:10].tolist(), "(b,q)", [(int(r) // Nv, int(r) % Nv) for r in rows[:10]], "n", rows.numel(), "cols", cols[:4].tolist(), "..", cols[-1].item(), "n", cols.numel())
print("bad iterations:", bad_total)
