import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mkg_analogy_amd import ops
ops.require_gpu()
M, N, K = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (100608, 2304, 768))]
cfg = int(sys.argv[4]) if len(sys.argv) > 4 else 256
A = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(5):
    ops.gemm_nt(A, W, out, tile_cfg=cfg)
torch.cuda.synchronize()
