"""gemm_tn (split-M, f32 atomics) vs torch.mm(X^T, Y) (hipBLASLt) on the weight-gradient shapes of the step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mkg_analogy_amd import ops
from tools.bench_kernels import timeit
ops.require_gpu()
DEV = "cuda"; BF = torch.bfloat16
for (M, NX, NY) in [(100608, 3072, 768), (100608, 768, 3072), (100608, 2304, 768), (100608, 768, 768), (16384, 3072, 768), (16384, 768, 768)]:
    X = (torch.randn(M, NX, device=DEV) * 0.1).to(BF); Y = (torch.randn(M, NY, device=DEV) * 0.1).to(BF)
    out = torch.zeros(NX, NY, device=DEV); cs = torch.zeros(NX, device=DEV)
    ms = timeit(lambda: ops.gemm_tn(X, Y, out, colsum=cs))
    ref = timeit(lambda: torch.mm(X.t(), Y))
    print(f"M={M} NX={NX} NY={NY}: gemm_tn {ms:.3f} ms {2*M*NX*NY/ms/1e9:.0f} TF/s | torch.mm(X^T,Y) bf16 out {ref:.3f} ms {2*M*NX*NY/ref/1e9:.0f} TF/s")
