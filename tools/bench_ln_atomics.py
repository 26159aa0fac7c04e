import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mkg_analogy_amd import ops
from tools.bench_kernels import timeit
ops.require_gpu()
DEV="cuda"; BF=torch.bfloat16; F32=torch.float32
for M in (256*393, 16384):
    H = 768
    x = torch.randn(M, H, device=DEV); g = torch.ones(H, device=DEV)
    mean = torch.zeros(M, device=DEV); rstd = torch.ones(M, device=DEV)
    dy = torch.randn(M, H, device=DEV).to(BF); ds = torch.empty(M, H, device=DEV); dsb = torch.empty(M, H, device=DEV, dtype=BF)
    dg = torch.zeros(H, device=DEV); db = torch.zeros(H, device=DEV)
    a = timeit(lambda: ops.ln_bwd(dy_bf16=dy, s=x, mean=mean, rstd=rstd, gamma=g, M=M, H=H, add_f32=x, ds_f32=ds, ds_bf16=dsb, dgamma=dg, dbeta=db, bf16_total=True))
    b = timeit(lambda: ops.ln_bwd(dy_bf16=dy, s=x, mean=mean, rstd=rstd, gamma=g, M=M, H=H, add_f32=x, ds_f32=ds, ds_bf16=dsb, bf16_total=True))
    print(f"ln_bwd M={M}: with dgamma/dbeta {a*1e3:.0f} us | without {b*1e3:.0f} us   ({M*H*16/a/1e6:.0f} / {M*H*16/b/1e6:.0f} GB/s)")
