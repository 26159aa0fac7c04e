import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mkg_analogy_amd import ops
ops.require_gpu()
B, S, Lp, nh, H = 256, 393, int(sys.argv[1]) if len(sys.argv) > 1 else 64, 12, 768
DEV = "cuda"; BF = torch.bfloat16
qkv = torch.randn(B * S, 3 * H, device=DEV).to(BF); tq = torch.randn(B * 64, 3 * H, device=DEV).to(BF)
ctx = torch.empty(B * S, H, device=DEV, dtype=BF); lse = torch.empty(B, nh, S, device=DEV)
kw = dict(q=qkv[:, :H], k=qkv[:, H:2*H], v=qkv[:, 2*H:], ctx=ctx, lse=lse, B=B, nh=nh, Sq=S, Sk=S, scale=0.125,
          pk=tq[:, H:2*H] if Lp else None, pv=tq[:, 2*H:] if Lp else None, Lp=Lp)
dctx = torch.randn(B * S, H, device=DEV).to(BF); dqkv = torch.empty(B * S, 3 * H, device=DEV, dtype=BF)
dt = torch.empty(B * 64, 3 * H, device=DEV, dtype=BF); delta = torch.empty(B, nh, S, device=DEV)
for _ in range(3):
    ops.attn_fwd(**kw)
    ops.attn_bwd(dctx=dctx, delta=delta, dq=dqkv[:, :H], dk=dqkv[:, H:2*H], dv=dqkv[:, 2*H:], dpk=dt[:, H:2*H] if Lp else None, dpv=dt[:, 2*H:] if Lp else None, **kw)
torch.cuda.synchronize()
