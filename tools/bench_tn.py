import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mkg_analogy_amd import ops
from tools.bench_kernels import timeit
ops.require_gpu()
DEV="cuda"; BF=torch.bfloat16
Mv = 256*393
for (M, NX, NY) in [(Mv, 2304, 768), (Mv, 768, 768), (Mv, 3072, 768), (Mv, 768, 3072), (16384, 3072, 768), (16384, 768, 768)]:
    X = (torch.randn(M, NX, device=DEV) * 0.1).to(BF); Y = torch.randn(M, NY, device=DEV).to(BF)
    out = torch.zeros(NX, NY, device=DEV); cs = torch.zeros(NX, device=DEV)
    tiles = ((NX+255)//256)*((NY+255)//256)
    for sp in (0, max(1, 128//tiles), max(1, 256//tiles), max(1, 384//tiles), max(1,512//tiles), max(1,1024//tiles)):
        ms = timeit(lambda: ops.gemm_tn(X, Y, out, colsum=cs, splits=sp))
        print(f"gemm_tn M={M} NX={NX} NY={NY} tiles={tiles} splits={sp}: {ms:.3f} ms {2*M*NX*NY/ms/1e9:.0f} TF/s")
