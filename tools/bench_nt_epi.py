"""Epilogue-heavy gemm_nt shapes of the step, 256x256 (1 WG/CU) vs 128x128 (2 WG/CU) tiles."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mkg_analogy_amd import ops
from tools.bench_kernels import timeit
ops.require_gpu()
DEV="cuda"; BF=torch.bfloat16
M = 256*393
cfgs = [int(c) for c in sys.argv[1:]] or [256, 128]
def run(name, N, K, mk):
    A = torch.randn(M, K, device=DEV).to(BF); W = (torch.randn(N, K, device=DEV) * 0.02).to(BF)
    kw, out = mk(N)
    for cfg in cfgs:
        ms = timeit(lambda: ops.gemm_nt(A, W, out, tile_cfg=cfg, **kw))
        print(f"{name:34s} N={N} K={K} cfg={cfg}: {ms:.3f} ms {2*M*N*K/ms/1e9:.0f} TF/s")
bias3072 = torch.randn(3072, device=DEV); bias768 = torch.randn(768, device=DEV)
run("fc1 fwd bias+act+act' -> bf16 x2", 3072, 768, lambda N: (dict(bias=bias3072, act=ops.ACT_QGELU, preact=torch.empty(M, N, device=DEV, dtype=BF), preact_grad=True), torch.empty(M, N, device=DEV, dtype=BF)))
run("fc2 dgrad * stored act' -> bf16", 3072, 768, lambda N: (dict(mulz=torch.randn(M, N, device=DEV).to(BF), mul_act=ops.ACT_STORED), torch.empty(M, N, device=DEV, dtype=BF)))
run("text fc1 fwd erf-gelu act+act' (M=16384)", 3072, 768, lambda N: (dict(bias=bias3072, act=ops.ACT_GELU, preact=torch.empty(M, N, device=DEV, dtype=BF), preact_grad=True, M=16384), torch.empty(M, N, device=DEV, dtype=BF)))
run("qkv fwd bias -> bf16", 2304, 768, lambda N: (dict(bias=torch.randn(N, device=DEV)), torch.empty(M, N, device=DEV, dtype=BF)))
run("out-proj bias+res_f32 -> f32", 768, 768, lambda N: (dict(bias=bias768, res_f32=torch.randn(M, N, device=DEV)), torch.empty(M, N, device=DEV)))
run("fc2 fwd bias+res_f32 -> f32", 768, 3072, lambda N: (dict(bias=bias768, res_f32=torch.randn(M, N, device=DEV)), torch.empty(M, N, device=DEV)))
run("plain -> bf16", 768, 768, lambda N: ({}, torch.empty(M, N, device=DEV, dtype=BF)))
run("plain -> bf16", 768, 3072, lambda N: ({}, torch.empty(M, N, device=DEV, dtype=BF)))
run("fc2 fwd bias+res_f32+C2 -> f32", 768, 3072, lambda N: (dict(bias=bias768, res_f32=torch.randn(M, N, device=DEV), C2=torch.empty(M, N, device=DEV, dtype=BF)), torch.empty(M, N, device=DEV)))
