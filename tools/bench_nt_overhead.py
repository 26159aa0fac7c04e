"""Per-tile fixed cost of gemm_nt: K=64 (one K-tile) vs K=768, plain bf16 epilogue / f32 out / residual."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mkg_analogy_amd import ops
from tools.bench_kernels import timeit
ops.require_gpu()
DEV="cuda"; BF=torch.bfloat16
M = 256*393
for N in (2304, 768):
    tiles = (M // 256) * (N // 256)
    for K in (64, 128, 256, 768):
        A = torch.randn(M, K, device=DEV).to(BF); W = (torch.randn(N, K, device=DEV) * 0.02).to(BF)
        out = torch.empty(M, N, device=DEV, dtype=BF)
        bias = torch.randn(N, device=DEV)
        res = torch.randn(M, N, device=DEV)
        outf = torch.empty(M, N, device=DEV)
        for name, fn in (("bf16", lambda: ops.gemm_nt(A, W, out, tile_cfg=256)),
                         ("bf16+bias", lambda: ops.gemm_nt(A, W, out, bias=bias, tile_cfg=256)),
                         ("f32+res+C2", lambda: ops.gemm_nt(A, W, outf, bias=bias, res_f32=res, C2=out, tile_cfg=256))):
            ms = timeit(fn)
            waves = tiles / 256
            print(f"N={N} K={K} {name}: {ms*1e3:.0f} us total, {ms*1e3/waves:.2f} us per tile-wave ({tiles} tiles)")
