import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mkg_analogy_amd import ops
from tools.bench_kernels import timeit
ops.require_gpu()
DEV="cuda"; BF=torch.bfloat16
Mv = 256*393
for (M, NX, NY) in [(Mv, 2304, 768), (Mv, 768, 768), (Mv, 3072, 768), (Mv, 768, 3072), (16384, 3072, 768)]:
    X = (torch.randn(M, NX, device=DEV) * 0.1).to(BF); Y = torch.randn(M, NY, device=DEV).to(BF)
    out = torch.zeros(NX, NY, device=DEV); cs = torch.zeros(NX, device=DEV)
    a = timeit(lambda: ops.gemm_tn(X, Y, out, colsum=cs)); b = timeit(lambda: ops.gemm_tn(X, Y, out))
    print(f"gemm_tn M={M} NX={NX} NY={NY}: with colsum {a:.3f} ms {2*M*NX*NY/a/1e9:.0f} TF/s | without {b:.3f} ms {2*M*NX*NY/b/1e9:.0f} TF/s")
