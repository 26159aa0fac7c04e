import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mkg_analogy_amd import ops
ops.require_gpu()
M, NX, NY = 100608, int(sys.argv[1]), int(sys.argv[2])
X = (torch.randn(M, NX, device="cuda") * 0.1).to(torch.bfloat16); Y = (torch.randn(M, NY, device="cuda") * 0.1).to(torch.bfloat16)
out = torch.zeros(NX, NY, device="cuda"); cs = torch.zeros(NX, device="cuda")
for _ in range(3):
    ops.gemm_tn(X, Y, out, colsum=cs)
torch.cuda.synchronize()
