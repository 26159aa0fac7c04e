"""Vision-stream LayerNorm backward at the bench shape (M = 256 x 393 rows of 768): the f32 gradient-stream form (16 bytes per element) against the bf16 form
(10 bytes per element, round 6), per-launch times through HIP events (incl. the dgamma / dbeta reduce launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mkg_analogy_amd import ops
from tools.bench_kernels import timeit
ops.require_gpu()
M, H = int(os.environ.get("M", 256 * 393)), 768
dev, F32, BF = "cuda", torch.float32, torch.bfloat16
xs = [torch.randn(M, H, device=dev) for _ in range(3)]
dys = [torch.randn(M, H, device=dev).to(BF) for _ in range(3)]
adds = [torch.randn(M, H, device=dev) for _ in range(3)]
addbs = [a.to(BF) for a in adds]
g = torch.randn(H, device=dev)
mean, rstd = torch.zeros(M, device=dev), torch.ones(M, device=dev)
ds, dsb = torch.empty(M, H, device=dev), torch.empty(M, H, device=dev, dtype=BF)
dg, db = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
it = [0]
def f32form():
    i = it[0] % 3; it[0] += 1
    ops.ln_bwd(dy_bf16=dys[i], s=xs[i], mean=mean, rstd=rstd, gamma=g, M=M, H=H, add_f32=adds[i], ds_f32=ds, ds_bf16=dsb, bf16_total=True, dgamma=dg, dbeta=db)
def bf16form():
    i = it[0] % 3; it[0] += 1
    ops.ln_bwd(dy_bf16=dys[i], s=xs[i], mean=mean, rstd=rstd, gamma=g, M=M, H=H, add_bf16=addbs[i], ds_bf16=dsb, bf16_total=True, dgamma=dg, dbeta=db)
for name, f, by in (("f32 gradient stream (16 B / element)", f32form, 16), ("bf16 gradient stream (10 B / element)", bf16form, 10)):
    ms = timeit(f)
    print(f"ln_bwd {name}: {ms * 1e3:.1f} us  {M * H * by / ms / 1e9:.2f} TB/s")
