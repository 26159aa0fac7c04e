"""P = 49 geometry (CLIP ViT-B/32: 99 vision tokens, M = 25 344 rows = 99 row tiles): 256x256 tiles against 128x128 tiles per product of the
vision layer, with the epilogue each one carries in the step (the N = 768 products are 297 tiles of 256x256 on 256 CUs = 1.16 rounds)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mkg_analogy_amd import ops
from tools.bench_kernels import timeit
ops.require_gpu()
DEV = "cuda"; BF = torch.bfloat16; F32 = torch.float32
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
Mv = B * 99
g = torch.Generator(device=DEV).manual_seed(0)
rn = lambda *s, dt=BF, sc=1.0: (torch.randn(*s, device=DEV, generator=g) * sc).to(dt)
cases = [("qkv fwd   [M,2304,768] bf16+bias", 2304, 768, {}),
         ("oproj fwd [M,768,768] f32+res", 768, 768, dict(res=True)),
         ("fc1 fwd   [M,3072,768] act+act'", 3072, 768, dict(fc1=True)),
         ("fc2 fwd   [M,768,3072] f32+res", 768, 3072, dict(res=True)),
         ("fc2 dgrad [M,3072,768] *act'", 3072, 768, dict(mulz=True)),
         ("fc1 dgrad [M,768,3072] bf16", 768, 3072, {}),
         ("oproj dgr [M,768,768] bf16", 768, 768, {}),
         ("qkv dgrad [M,768,2304] bf16", 768, 2304, {})]
tot = {128: 0.0, 256: 0.0, 0: 0.0, "best": 0.0}
for name, N, K, kw in cases:
    A = [rn(Mv, K) for _ in range(6)]; W = rn(N, K, sc=0.02); bias = rn(N, dt=F32)
    res = rn(Mv, N, dt=F32) if kw.get("res") else None
    z = rn(Mv, N) if kw.get("mulz") else None
    out = torch.empty(Mv, N, device=DEV, dtype=F32 if kw.get("res") else BF)
    pre = torch.empty(Mv, N, device=DEV, dtype=BF) if kw.get("fc1") else None
    r = {}
    for cfg in (256, 128, 0):        # 0 = the dispatcher's choice (round 5: one round of 256x256 tiles + 128x128 tiles over the remaining rows)
        it = [0]
        def f():
            a = A[it[0] % 6]; it[0] += 1
            if kw.get("res"): ops.gemm_nt(a, W, out, bias=bias, res_f32=res, tile_cfg=cfg)
            elif kw.get("fc1"): ops.gemm_nt(a, W, out, bias=bias, act=ops.ACT_QGELU, preact=pre, preact_grad=True, tile_cfg=cfg)
            elif kw.get("mulz"): ops.gemm_nt(a, W, out, mulz=z, mul_act=ops.ACT_STORED, tile_cfg=cfg)
            else: ops.gemm_nt(a, W, out, bias=bias, tile_cfg=cfg)
        r[cfg] = timeit(f)
        tot[cfg] += r[cfg]
    tot["best"] += min(r[256], r[128])
    t256 = ((Mv + 255) // 256) * ((N + 255) // 256)
    print(f"{name:36s} tiles256 {t256:5d} ({t256 / 256:.2f} rounds)   256: {r[256]:.4f} ms {2 * Mv * N * K / r[256] / 1e9:6.0f} TF/s   128: {r[128]:.4f} ms {2 * Mv * N * K / r[128] / 1e9:6.0f} TF/s   x{r[256] / r[128]:.3f}   auto: {r[0]:.4f} ms {2 * Mv * N * K / r[0] / 1e9:6.0f} TF/s")
print(f"sum per layer: 256 {tot[256]:.4f} ms, 128 {tot[128]:.4f} ms, best-of {tot['best']:.4f} ms, auto {tot[0]:.4f} ms")
