"""Does the order in which a bandwidth-bound consumer sweeps its input matter?  (MI355X: 256 MiB memory-side Infinity Cache.)

The vision stream's LayerNorm kernels read [M, 768] f32 matrices (309 MB at M = 100 608) that the kernel in front of them has just
written, front to back, and sweep them front to back themselves: with an LRU-like memory-side cache that is the worst order -- the rows
still cached when the kernel starts are the LAST ones, and by the time the sweep reaches them its own traffic has evicted them.
``MART_LN_REV`` (bit 0: forward kernel, bit 1: backward kernel) makes the fast LayerNorm kernels sweep from the last row to the first.
This probe times the kernel behind its real producer in both orders, plus a cold control (1 GiB of unrelated traffic in between).

The library reads the knob once per process, so the probe re-runs itself per setting.

usage (GPU box): python tools/mall_probe.py [reps]
"""
import os, subprocess, sys
if "MALL_PROBE_CHILD" not in os.environ:
    for rev in ("0", "3"):
        subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=dict(os.environ, MART_LN_REV=rev, MALL_PROBE_CHILD="1"), check=True)
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mkg_analogy_amd import ops

ops.require_gpu()
dev = torch.device("cuda", 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, Nv, H = 256, 393, 768
M = B * Nv
BF, F32 = torch.bfloat16, torch.float32
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s, dt=F32, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(dt)

ctx, w_o, b_o, xv = rn(M, H, dt=BF), rn(H, H, dt=BF, sc=0.02), rn(H), rn(M, H)
gamma, beta = 1.0 + 0.1 * rn(H), 0.1 * rn(H)
x1, h2, m2, r2 = torch.empty(M, H, device=dev), torch.empty(M, H, device=dev, dtype=BF), torch.empty(M, device=dev), torch.empty(M, device=dev)
scrub_a, scrub_b = torch.empty(1 << 28, device=dev), torch.empty(1 << 28, device=dev)          # 1 GiB each

dqkv, w_qkv_t = rn(M, 3 * H, dt=BF, sc=0.1), rn(H, 3 * H, dt=BF, sc=0.02)
dh1, dres = torch.empty(M, H, device=dev, dtype=BF), rn(M, H, sc=0.1)
dxv, dxvb = torch.empty(M, H, device=dev), torch.empty(M, H, device=dev, dtype=BF)
dg, db = torch.zeros(H, device=dev), torch.zeros(H, device=dev)


def producer_fwd():
    ops.gemm_nt(ctx, w_o, x1, bias=b_o, res_f32=xv)                       # out-proj + f32 residual: writes x1 (309 MB)


def consumer_fwd():
    ops.ln_fwd(x_f32=x1, gamma=gamma, beta=beta, eps=1e-5, M=M, H=H, mean=m2, rstd=r2, out_bf16=h2)


def producer_bwd():
    ops.gemm_nt(dqkv, w_qkv_t, dh1)                                       # QKV data gradient: writes dh1 (155 MB)


def consumer_bwd():
    ops.ln_bwd(dy_bf16=dh1, s=x1, mean=m2, rstd=r2, gamma=gamma, M=M, H=H, add_f32=dres, ds_f32=dxv, ds_bf16=dxvb, bf16_total=True, dgamma=dg, dbeta=db)


def timed(producer, consumer, scrub):
    ts = []
    for _ in range(reps):
        producer()
        if scrub:
            scrub_b.copy_(scrub_a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); consumer(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


producer_fwd(); consumer_fwd(); producer_bwd(); consumer_bwd(); torch.cuda.synchronize()
rev = int(os.environ.get("MART_LN_REV", "0"))
print(f"MART_LN_REV={rev}: checksums h2 {float(h2.float().sum()):.6f} mean {float(m2.sum()):.6f} dxv {float(dxv.double().sum()):.9f}")
for name, prod, cons, mb in (("ln_fwd_fast_k (x f32 309 MB in, bf16 155 MB out)", producer_fwd, consumer_fwd, 464.0),
                             ("ln_bwd_fast_k (dy bf16 + x f32 + residual gradient f32 in, f32 + bf16 out)", producer_bwd, consumer_bwd, 1237.0)):
    print(name)
    for scrub in (False, True):
        med, best = timed(prod, cons, scrub)
        print(f"  sweep {'last row first' if rev else 'first row first'}, {'cold (1 GiB copy in between)' if scrub else 'right behind its producer'}: "
              f"median {med * 1e3:.1f} us ({mb / med / 1e3:.2f} TB/s), best {best * 1e3:.1f} us")
