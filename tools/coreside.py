"""Do memory-bound kernels co-reside with the 256 x 256 GEMM workgroups on a CU?  (round 4, docs/LAB_r01-r05.md section 4.4)

A GEMM workgroup holds 128 KB of the CU's 160 KB LDS and 8 waves x 221-256 VGPRs: at 224 allocated registers per wave a SIMD has 64 registers per
lane (and the CU 32 KB of LDS) left, so a wave of a <= 64-VGPR kernel fits next to it; at 256 nothing does.  This script times a GEMM loop on one
stream and a streaming kernel loop on another, alone and together:  together ~ max(alone) = the kernels share CUs;  together ~ sum = they take turns.

usage: python tools/coreside.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mkg_analogy_amd import ops  # noqa: E402

ops.require_gpu()
dev = torch.device("cuda", 0)
BF, F32 = torch.bfloat16, torch.float32
M, H, I = 256 * 394, 768, 3072
g = torch.Generator(device=dev).manual_seed(1)
rn = lambda *s, dt=BF, sc=0.1: (torch.randn(*s, device=dev, generator=g) * sc).to(dt)

A, Wq, Cq = rn(M, H), rn(3 * H, H), torch.empty(M, 3 * H, device=dev, dtype=BF)          # QKV: persistent kernel, 253 VGPRs
Wo, Co, Rres = rn(H, H), torch.empty(M, H, device=dev, dtype=F32), rn(M, H, dt=F32)        # out-proj: f32 + residual, 231 VGPRs
X, Y, dW = rn(M, I), rn(M, H), torch.zeros(I, H, device=dev)                               # fc1 weight gradient: gemm_tn8, 224 VGPRs
src = rn(M * H, dt=F32)
dst = torch.empty(M * H, device=dev, dtype=BF)
xln = rn(M, H, dt=F32, sc=1.0)
gam, bet = torch.ones(H, device=dev), torch.zeros(H, device=dev)
mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
hln = torch.empty(M, H, device=dev, dtype=BF)
dy = rn(M, H)
gres = rn(M, H, dt=F32)
dx, dxb = torch.empty(M, H, device=dev), torch.empty(M, H, device=dev, dtype=BF)
dg, db = torch.zeros(H, device=dev), torch.zeros(H, device=dev)

gemms = {
    "gemm_nt QKV (persistent, 253 VGPR)": lambda: ops.gemm_nt(A, Wq, Cq),
    "gemm_nt out-proj f32+res (231 VGPR)": lambda: ops.gemm_nt(A, Wo, Co, res_f32=Rres),
    "gemm_tn8 fc1 wgrad (224 VGPR)": lambda: ops.gemm_tn(X, Y, dW),
}
streams_k = {
    "cast_f32_bf16 (14 VGPR)": lambda: ops.cast_f32_bf16(src, dst),
    "ln_fwd_fast (98 VGPR)": lambda: ops.ln_fwd(x_f32=xln, gamma=gam, beta=bet, eps=1e-5, M=M, H=H, mean=mean, rstd=rstd, out_bf16=hln),
    "ln_bwd_fast (183 VGPR)": lambda: ops.ln_bwd(dy_bf16=dy, s=xln, mean=mean, rstd=rstd, gamma=gam, M=M, H=H, add_f32=gres, ds_f32=dx, ds_bf16=dxb,
                                                 bf16_total=True, dgamma=dg, dbeta=db),
}
ops.ln_fwd(x_f32=xln, gamma=gam, beta=bet, eps=1e-5, M=M, H=H, mean=mean, rstd=rstd, out_bf16=hln)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(f1, n1, f2, n2):
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    s1.wait_event(e0); s2.wait_event(e0)
    if f1:
        with torch.cuda.stream(s1):
            for _ in range(n1):
                f1()
            e1.record()
    if f2:
        with torch.cuda.stream(s2):
            for _ in range(n2):
                f2()
            e2.record()
    torch.cuda.synchronize()
    return (e0.elapsed_time(e1) if f1 else 0.0), (e0.elapsed_time(e2) if f2 else 0.0)


for f in list(gemms.values()) + list(streams_k.values()):
    for _ in range(3):
        f()
torch.cuda.synchronize()
print(f"{'GEMM loop':40s} {'streaming loop':26s} {'alone ms (g / s)':>20s} {'together ms (g / s)':>22s}   together / sum   together / max")
for gn, gf in gemms.items():
    tg = min(run(gf, 20, None, 0)[0] for _ in range(3))
    for sn, sf in streams_k.items():
        ts1 = min(run(None, 0, sf, 20)[1] for _ in range(3)) / 20
        n2 = max(1, int(round(tg / ts1)))                       # equal alone-durations
        ts = min(run(None, 0, sf, n2)[1] for _ in range(3))
        both = min((max(run(gf, 20, sf, n2)) for _ in range(3)))
        b = run(gf, 20, sf, n2)
        print(f"{gn:40s} {sn:26s} {tg:9.2f} / {ts:8.2f} {b[0]:11.2f} / {b[1]:8.2f}   {both / (tg + ts):10.2f}   {both / max(tg, ts):14.2f}")
