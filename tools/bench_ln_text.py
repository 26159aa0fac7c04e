"""Text-stream LayerNorm (dropout + residual + LayerNorm, BertSelfOutput / BertOutput; M = 256 x 64 rows of 768) forward / backward at the bench
shape, per-launch times through HIP events; the memory floor is 197 MB per launch (~33 us at 6 TB/s)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mkg_analogy_amd import ops
from tools.bench_kernels import timeit
ops.require_gpu()
M, H = int(os.environ.get("M", 256 * 64)), 768
F32, BF = torch.float32, torch.bfloat16
dev = "cuda"
x = torch.randn(M, H, device=dev); so = torch.randn(M, H, device=dev).to(BF)
g, b = torch.randn(H, device=dev), torch.randn(H, device=dev)
mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
s1, a, ab = torch.empty(M, H, device=dev), torch.empty(M, H, device=dev), torch.empty(M, H, device=dev, dtype=BF)
for p in (0.1, 0.0):
    f = lambda: ops.ln_fwd(x_f32=x, y_bf16=so, p_drop=p, seed=7, gamma=g, beta=b, eps=1e-12, M=M, H=H, mean=mean, rstd=rstd, s_out=s1, out_f32=a, out_bf16=ab)
    print(f"text ln fwd p={p}: {timeit(f) * 1e3:.1f} us")
    d32, d16 = torch.randn(M, H, device=dev), torch.randn(M, H, device=dev).to(BF)
    ds, dsb = torch.empty(M, H, device=dev), torch.empty(M, H, device=dev, dtype=BF)
    dg, db = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    f2 = lambda: ops.ln_bwd(dy_f32=d32, dy_bf16=d16, s=s1, mean=mean, rstd=rstd, gamma=g, M=M, H=H, ds_f32=ds, ds_bf16=dsb, p_drop=p, seed=7, dgamma=dg, dbeta=db)
    print(f"text ln bwd p={p}: {timeit(f2) * 1e3:.1f} us  (incl. the dgamma / dbeta reduce launch)")
