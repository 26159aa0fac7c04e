import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mkg_analogy_amd import ops
from tools.bench_kernels import timeit
ops.require_gpu()
DEV="cuda"; BF=torch.bfloat16
Mv = 256*393
for (M, N, K) in [(Mv, 2304, 768), (Mv, 768, 3072), (Mv, 3072, 768)]:
    A = torch.randn(M, K, device=DEV).to(BF); W = (torch.randn(N, K, device=DEV) * 0.02).to(BF)
    out = torch.empty(M, N, device=DEV, dtype=BF)
    for cfg in (256, 999):
        ms = timeit(lambda: ops.gemm_nt(A, W, out, tile_cfg=cfg))
        print(f"gemm_nt M={M} N={N} K={K} cfg={cfg}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.0f} TF/s")
