import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mkg_analogy_amd import ops
from tools.bench_kernels import timeit
ops.require_gpu()
DEV="cuda"; BF=torch.bfloat16
M = 256*393; N = 2304
tiles = (M // 256) * (N // 256)
for K in (64, 768):
    A = torch.randn(M, K, device=DEV).to(BF); W = (torch.randn(N, K, device=DEV) * 0.02).to(BF)
    out = torch.empty(M, N, device=DEV, dtype=BF)
    for cfg, name in ((256, "full"), (9993, "LDS staging, no global stores"), (9992, "no epilogue")):
        ms = timeit(lambda: ops.gemm_nt(A, W, out, tile_cfg=cfg))
        print(f"K={K} {name}: {ms*1e3:.0f} us total, {ms*1e3*256/tiles:.2f} us per tile-wave")
