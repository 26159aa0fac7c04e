"""Text attention (BertSelfAttention, 64 x 64 per head, mask + adaptive reweight + dropout) forward / backward at the bench shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mkg_analogy_amd import ops
from tools.bench_kernels import timeit
ops.require_gpu()
B, L, nh, H = 256, int(os.environ.get("L", 64)), 12, 768
BF = torch.bfloat16
qkv = torch.randn(B * L, 3 * H, device="cuda").to(BF)
dctx = torch.randn(B * L, H, device="cuda").to(BF)
ctx = torch.empty(B * L, H, device="cuda", dtype=BF)
lse = torch.empty(B, nh, L, device="cuda")
delta = torch.empty(B, nh, L, device="cuda")
dqkv = torch.empty(B * L, 3 * H, device="cuda", dtype=BF)
am = torch.ones(B, L, device="cuda", dtype=torch.int64); am[:, L - 14:] = 0
sep = torch.full((B, 6), 20, device="cuda", dtype=torch.int64)
w0, w1 = torch.tensor([0.25], device="cuda"), torch.tensor([0.5], device="cuda")
dw = torch.zeros(2, device="cuda")
kw = dict(q=qkv[:, :H], k=qkv[:, H:2*H], v=qkv[:, 2*H:], ctx=ctx, lse=lse, B=B, nh=nh, Sq=L, Sk=L, scale=0.125, attn_mask=am, sep=sep[:, 2:], sep_stride=6,
          w0=w0, w1=w1, p_drop=float(os.environ.get("P", 0.1)), seed=1234)
print(f"text attn fwd: {timeit(lambda: ops.attn_fwd(**kw)) * 1e3:.1f} us")
ops.attn_fwd(**kw)
print(f"text attn bwd (dq + dw reduce + dkv): {timeit(lambda: ops.attn_bwd(dctx=dctx, delta=delta, dq=dqkv[:, :H], dk=dqkv[:, H:2*H], dv=dqkv[:, 2*H:], dw=dw, **kw)) * 1e3:.1f} us")
