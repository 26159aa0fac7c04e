import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mkg_analogy_amd import ops
from tools.bench_kernels import timeit
ops.require_gpu()
DEV="cuda"; BF=torch.bfloat16
Mv = 256*393
cfgs = [int(c) for c in sys.argv[1:]] or [256, 2560]
for (M, N, K) in [(Mv, 2304, 768), (Mv, 768, 768), (Mv, 3072, 768), (Mv, 768, 3072), (Mv, 768, 2304), (16384, 3072, 768)]:
    A = torch.randn(M, K, device=DEV).to(BF); W = (torch.randn(N, K, device=DEV) * 0.02).to(BF)
    out = torch.empty(M, N, device=DEV, dtype=BF); ref = torch.empty(M, N, device=DEV, dtype=BF)
    ops.gemm_nt(A, W, ref, tile_cfg=256)
    for cfg in cfgs:
        ops.gemm_nt(A, W, out, tile_cfg=cfg)
        err = (out.float() - ref.float()).abs().max().item()
        ms = timeit(lambda: ops.gemm_nt(A, W, out, tile_cfg=cfg))
        print(f"gemm_nt M={M} N={N} K={K} cfg={cfg}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.0f} TF/s  maxdiff_vs_256 {err:.3g}")
