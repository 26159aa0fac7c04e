"""Text attention backward: fused 64 x 64 kernel vs the two-pass kernels on the same inputs (two processes: MART_ATTN_TEXT_FUSED is read once).
    MART_ATTN_TEXT_FUSED=0 python tools/cmp_text_bwd.py save /tmp/a.pt; MART_ATTN_TEXT_FUSED=1 python tools/cmp_text_bwd.py save /tmp/b.pt
    python tools/cmp_text_bwd.py cmp /tmp/a.pt /tmp/b.pt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

if sys.argv[1] == "cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        x, y = a[k].float(), b[k].float()
        d = (x - y).abs()
        print(f"{k:6s} max|a| {float(x.abs().max()):.4e}  max|a-b| {float(d.max()):.4e}  rms(a-b)/rms(a) {float(d.pow(2).mean().sqrt() / x.pow(2).mean().sqrt()):.3e}  "
              f"elements differing {int((d > 0).sum())} / {d.numel()}")
    sys.exit(0)

from mkg_analogy_amd import ops
ops.require_gpu()
B, L, nh, H = 64, 64, 12, 768
BF = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(5)
qkv = torch.randn(B * L, 3 * H, device="cuda", generator=g).to(BF)
dctx = (torch.randn(B * L, H, device="cuda", generator=g) * 0.1).to(BF)
pre = (torch.randn(B * L, 3 * H, device="cuda", generator=g) * 0.05).to(BF)
ctx = torch.empty(B * L, H, device="cuda", dtype=BF)
lse = torch.empty(B, nh, L, device="cuda")
delta = torch.empty(B, nh, L, device="cuda")
am = torch.ones(B, L, device="cuda", dtype=torch.int64)
am[:, 50:] = 0
am[1, 33:] = 0
sep = torch.full((B, 6), 20, device="cuda", dtype=torch.int64)
sep[:, 2] = torch.arange(B, device="cuda") % 40 + 5
w0, w1 = torch.tensor([0.25], device="cuda"), torch.tensor([0.75], device="cuda")
out = {}
for acc in (False, True):
    dqkv = pre.clone()
    dw = torch.zeros(2, device="cuda")
    kw = dict(q=qkv[:, :H], k=qkv[:, H:2*H], v=qkv[:, 2*H:], ctx=ctx, lse=lse, B=B, nh=nh, Sq=L, Sk=L, scale=0.125, attn_mask=am, sep=sep[:, 2:], sep_stride=6,
              w0=w0, w1=w1, p_drop=0.1, seed=1234)
    ops.attn_fwd(**kw)
    ops.attn_bwd(dctx=dctx, delta=delta, dq=dqkv[:, :H], dk=dqkv[:, H:2*H], dv=dqkv[:, 2*H:], dw=dw, accum_dkv=acc, **kw)
    torch.cuda.synchronize()
    t = "acc" if acc else "new"
    out["dq_" + t], out["dk_" + t], out["dv_" + t], out["dw_" + t] = dqkv[:, :H].cpu(), dqkv[:, H:2*H].cpu(), dqkv[:, 2*H:].cpu(), dw.cpu()
    out["delta_" + t] = delta.cpu()
torch.save(out, sys.argv[2])
