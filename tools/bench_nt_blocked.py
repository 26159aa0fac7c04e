import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mkg_analogy_amd import ops
from tools.bench_kernels import timeit
ops.require_gpu()
DEV="cuda"; BF=torch.bfloat16
Mv = 256*393
for (M, N, K) in [(Mv, 2304, 768), (Mv, 768, 768), (Mv, 3072, 768), (Mv, 768, 3072), (Mv, 768, 2304), (16384, 3072, 768)]:
    A = torch.randn(M, K, device=DEV).to(BF); W = (torch.randn(N, K, device=DEV) * 0.02).to(BF)
    Wb = torch.empty_like(W)
    table = torch.tensor([[0, 0, N, K]], dtype=torch.int64, device=DEV)
    ops.block_table(W, Wb, table, 1)
    ref_blk = W.view(N // 256, 256, K // 64, 64).permute(0, 2, 1, 3).contiguous().view(-1)
    assert torch.equal(Wb.view(-1), ref_blk), "block_table layout"
    out = torch.empty(M, N, device=DEV, dtype=BF); ref = torch.empty(M, N, device=DEV, dtype=BF)
    ops.gemm_nt(A, W, ref, tile_cfg=256)
    ops.gemm_nt(A, Wb.view(N, K), out, tile_cfg=256, b_blocked=True)
    err = (out.float() - ref.float()).abs().max().item()
    m0 = timeit(lambda: ops.gemm_nt(A, W, ref, tile_cfg=256))
    m1 = timeit(lambda: ops.gemm_nt(A, Wb.view(N, K), out, tile_cfg=256, b_blocked=True))
    print(f"gemm_nt M={M} N={N} K={K}: row-major {m0:.3f} ms {2*M*N*K/m0/1e9:.0f} TF/s | blocked-B {m1:.3f} ms {2*M*N*K/m1/1e9:.0f} TF/s  maxdiff {err:.3g}")
