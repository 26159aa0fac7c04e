"""Package power and shader clock per kernel family: each family runs alone in a 3 s loop while rocm-smi is sampled at ~4 Hz (round 4, docs/LAB_r01-r05.md section 4.1d).
usage: python tools/power_kernels.py"""
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mkg_analogy_amd import ops  # noqa: E402

ops.require_gpu()
dev = torch.device("cuda", 0)
BF, F32 = torch.bfloat16, torch.float32
B, S, nh, H, I = 256, 393, 12, 768, 3072
M = B * S
g = torch.Generator(device=dev).manual_seed(1)
rn = lambda *s, dt=BF, sc=0.1: (torch.randn(*s, device=dev, generator=g) * sc).to(dt)
A, Wq, Cq = rn(M, H), rn(3 * H, H), torch.empty(M, 3 * H, device=dev, dtype=BF)
Wz = torch.zeros(3 * H, H, device=dev, dtype=BF); Az = torch.zeros(M, H, device=dev, dtype=BF)
X, Y, dW = rn(M, I), rn(M, H), torch.zeros(I, H, device=dev)
qkv = rn(M, 3 * H, sc=1.0); ctx = torch.empty(M, H, device=dev, dtype=BF); lse = torch.empty(B, nh, S, device=dev)
dctx = rn(M, H); dqkv = torch.empty(M, 3 * H, device=dev, dtype=BF); delta = torch.empty(B, nh, S, device=dev)
xln = rn(M, H, dt=F32, sc=1.0); gam, bet = torch.ones(H, device=dev), torch.zeros(H, device=dev)
mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev); hln = torch.empty(M, H, device=dev, dtype=BF)
gres = rn(M, H, dt=F32); dx, dxb = torch.empty(M, H, device=dev), torch.empty(M, H, device=dev, dtype=BF)
dg, db = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
akw = dict(q=qkv[:, :H], k=qkv[:, H:2 * H], v=qkv[:, 2 * H:], ctx=ctx, lse=lse, B=B, nh=nh, Sq=S, Sk=S, scale=0.125)
ops.ln_fwd(x_f32=xln, gamma=gam, beta=bet, eps=1e-5, M=M, H=H, mean=mean, rstd=rstd, out_bf16=hln)
ops.attn_fwd(**akw)
fams = {
    "gemm_nt QKV [M,2304,768] (random operands)": lambda: ops.gemm_nt(A, Wq, Cq),
    "gemm_nt QKV, zero operands": lambda: ops.gemm_nt(Az, Wz, Cq),
    "gemm_tn8 fc1 weight gradient": lambda: ops.gemm_tn(X, Y, dW),
    "attn_fwd_k vision": lambda: ops.attn_fwd(**akw),
    "attn_bwd_fused_k vision": lambda: ops.attn_bwd(dctx=dctx, delta=delta, dq=dqkv[:, :H], dk=dqkv[:, H:2 * H], dv=dqkv[:, 2 * H:], **akw),
    "ln_fwd_fast_k": lambda: ops.ln_fwd(x_f32=xln, gamma=gam, beta=bet, eps=1e-5, M=M, H=H, mean=mean, rstd=rstd, out_bf16=hln),
    "ln_bwd_fast_k": lambda: ops.ln_bwd(dy_bf16=dctx, s=xln, mean=mean, rstd=rstd, gamma=gam, M=M, H=H, add_f32=gres, ds_f32=dx, ds_bf16=dxb,
                                        bf16_total=True, dgamma=dg, dbeta=db),
}
samples = []
stop = False


def sampler():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
        p = re.search(r"Package Power \(W\): ([0-9.]+)", out); c = re.search(r"sclk clock level: \S+: \((\d+)Mhz\)", out)
        if p and c:
            samples.append((time.perf_counter(), float(p.group(1)), int(c.group(1))))


th = threading.Thread(target=sampler); th.start()
print(f"{'kernel family (alone, 3 s loop)':48s} {'ms / launch':>11s} {'W (median)':>10s} {'sclk MHz':>9s}")
for name, f in fams.items():
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 3.0:
        for _ in range(20):
            f()
        n += 20
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    s = [x for x in samples if t0 + 1.0 < x[0] < t1]
    pw = sorted(x[1] for x in s); ck = sorted(x[2] for x in s)
    print(f"{name:48s} {(t1 - t0) / n * 1e3:11.4f} {pw[len(pw) // 2] if pw else -1:10.0f} {ck[len(ck) // 2] if ck else -1:9d}   ({len(s)} samples)")
    time.sleep(1.0)
stop = True; th.join()
