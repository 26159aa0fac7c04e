"""vision_assemble_bwd / text_embed_scatter / transposes at the bench shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mkg_analogy_amd import ops
from tools.bench_kernels import timeit
ops.require_gpu()
DEV = "cuda"; BF = torch.bfloat16
B, P, H, L, V = 256, 196, 768, 64, 42007
ds = torch.randn(B * (1 + 2 * P), H, device=DEV)
dpatch = torch.empty(B * 2 * P, H, device=DEV, dtype=BF)
dcls = torch.zeros(H, device=DEV); dpos = torch.zeros(P + 1, H, device=DEV)
ms = timeit(lambda: ops.vision_assemble_bwd(ds, dpatch, dcls, dpos, B, P, H))
print(f"vision_assemble_bwd: {ms:.3f} ms  {(ds.numel()*4 + dpatch.numel()*2)/ms/1e9:.2f} TB/s")
dse = torch.randn(B * L, H, device=DEV)
ids = torch.randint(0, V, (B, L), device=DEV); tt = torch.randint(0, 2, (B, L), device=DEV)
dword = torch.zeros(V, H, device=DEV); dp = torch.zeros(512, H, device=DEV); dt = torch.zeros(2, H, device=DEV)
ms = timeit(lambda: ops.text_embed_scatter(dse, ids, tt, dword, dp, dt, B, L, H))
print(f"text_embed_scatter: {ms:.3f} ms")
x = torch.randn(B * 393, 768, device=DEV).to(BF); xt = torch.empty(B, 768, 448, device=DEV, dtype=BF)
ms = timeit(lambda: ops.transpose_bf16(x, xt, 393, 768, 448, batch=B, stride_i=393 * 768, stride_o=768 * 448))
print(f"transpose_bf16 [256 x 393 x 768]: {ms:.3f} ms  {2*x.numel()*2/ms/1e9:.2f} TB/s")
