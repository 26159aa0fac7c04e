"""fc1-forward GEMM with the same output buffers every launch vs rotating through NB distinct ones (as a step does)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mkg_analogy_amd import ops
from tools.bench_kernels import timeit
ops.require_gpu()
DEV = "cuda"; BF = torch.bfloat16
M, N, K = 256 * 393, 3072, 768
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 12
W = (torch.randn(N, K, device=DEV) * 0.02).to(BF)
bias = torch.randn(N, device=DEV)
As = [torch.randn(M, K, device=DEV).to(BF) for _ in range(NB)]
outs = [torch.empty(M, N, device=DEV, dtype=BF) for _ in range(NB)]
pres = [torch.empty(M, N, device=DEV, dtype=BF) for _ in range(NB)]
for name, rot_a, rot_o in (("same A, same out", 0, 0), ("rotating A", 1, 0), ("rotating out", 0, 1), ("rotating both", 1, 1), ("same A, same out", 0, 0)):
    i = [0]
    def f():
        k = i[0] % NB; i[0] += 1
        ops.gemm_nt(As[k * rot_a], W, outs[k * rot_o], bias=bias, act=ops.ACT_QGELU, preact=pres[k * rot_o], preact_grad=True)
    ms = timeit(f, iters=48)
    print(f"fc1 fwd {name:18s}: {ms:.3f} ms {2*M*N*K/ms/1e9:.0f} TF/s")
    def g():
        k = i[0] % NB; i[0] += 1
        ops.gemm_nt(As[k * rot_a], W, outs[k * rot_o], bias=bias)
    ms = timeit(g, iters=48)
    print(f"bias    {name:18s}: {ms:.3f} ms {2*M*N*K/ms/1e9:.0f} TF/s")
