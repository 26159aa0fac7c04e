"""Cycle stamps (s_memtime, 100 MHz constant clock on gfx950 -> converted with the measured shader clock is NOT attempted: stamps are printed in
REFCLK ticks and as shares) of one vision-attention workgroup, forward and fused backward, in the middle of a bench-shape launch.
Needs a variant library: tools/build_variant.sh attention.hip /tmp/attn_stamps.so -DATTN_STAMPS ; MART_HIP_LIB=/tmp/attn_stamps.so python tools/attn_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mkg_analogy_amd import ops, _lib

ops.require_gpu()
lib = ctypes.CDLL(os.environ["MART_HIP_LIB"])
DEV, BF = "cuda", torch.bfloat16
B, nh, H, S = 256, 12, 768, 393
for Lp in (0, 64):
    qkv = torch.randn(B * S, 3 * H, device=DEV).to(BF)
    tq = torch.randn(B * 64, 3 * H, device=DEV).to(BF)
    dctx = torch.randn(B * S, H, device=DEV).to(BF)
    ctx = torch.empty(B * S, H, device=DEV, dtype=BF)
    lse = torch.empty(B, nh, S, device=DEV)
    dqkv = torch.empty(B * S, 3 * H, device=DEV, dtype=BF)
    dt = torch.empty(B * 64, 3 * H, device=DEV, dtype=BF)
    delta = torch.empty(B, nh, S, device=DEV)
    kw = dict(q=qkv[:, :H], k=qkv[:, H:2*H], v=qkv[:, 2*H:], ctx=ctx, lse=lse, B=B, nh=nh, Sq=S, Sk=S, scale=0.125,
              pk=tq[:, H:2*H] if Lp else None, pv=tq[:, 2*H:] if Lp else None, Lp=Lp)
    for _ in range(3):
        ops.attn_fwd(**kw)
        ops.attn_bwd(dctx=dctx, delta=delta, dq=dqkv[:, :H], dk=dqkv[:, H:2*H], dv=dqkv[:, 2*H:],
                     dpk=dt[:, H:2*H] if Lp else None, dpv=dt[:, 2*H:] if Lp else None, **kw)
    torch.cuda.synchronize()
    out = np.zeros(2 * 8 * 32, dtype=np.uint64)
    assert lib.mart_debug_attn_stamps(out.ctypes.data_as(ctypes.c_void_p)) == 0
    st = out.reshape(2, 8, 32).astype(np.int64)
    ntk = (S + Lp + 63) // 64
    print(f"=== forward, keys {S + Lp} ({ntk} tiles), workgroup part 1 of (b={B//2}, h=5): ticks since wave 0's entry stamp")
    t0 = st[0, 0, 0]
    for w in range(4):
        r = st[0, w, :5 + ntk] - t0
        print(f" wave {w}: entry {r[0]:6d} | prologue done {r[1]:6d} | tiles " + " ".join(f"{r[2 + k]:6d}" for k in range(ntk)) +
              f" | loop end {r[2 + ntk]:6d} | sync {r[3 + ntk]:6d} | end {r[4 + ntk]:6d}")
    r = st[0, 0, :5 + ntk] - t0
    tot = r[4 + ntk]
    print(f" wave 0 shares: prologue {100 * r[1] / tot:.1f} %, loop {100 * (r[2 + ntk] - r[1]) / tot:.1f} % ({(r[2 + ntk] - r[2]) / ntk:.0f} ticks per tile), epilogue {100 * (tot - r[2 + ntk]) / tot:.1f} %")
    nt = (S + 31) // 32
    print(f"=== fused backward, {nt} query tiles")
    t0 = st[1, 0, 0]
    for w in range(8):
        r = st[1, w, :6 + nt] - t0
        print(f" wave {w}: entry {r[0]:6d} | prologue issued {r[1]:6d} | barrier {r[2]:6d} | iterations " + " ".join(f"{r[3 + k]:6d}" for k in range(nt + 1)) +
              f" | epi sync {r[4 + nt]:6d} | end {r[5 + nt]:6d}")
    print(" iteration 5 in detail (ticks since its loop top): staged | dq (waves 0-3) | kb0: S+dP, a0, a1 | kb1: S+dP, a0, a1 | s_tile end | dq (waves 4-7) | (barrier = next loop top)")
    for w in range(8):
        f = st[1, w, 20:31] - st[1, w, 20]
        print(f" wave {w}: " + " ".join(f"{x:6d}" for x in f[1:]) + f" | {st[1, w, 3 + 5] - st[1, w, 20]:6d}")
    r = st[1, 0, :6 + nt] - t0
    tot = r[5 + nt]
    print(f" wave 0 shares: prologue {100 * r[2] / tot:.1f} %, loop {100 * (r[3 + nt] - r[2]) / tot:.1f} % ({(r[3 + nt - 1] - r[3]) / (nt - 1):.0f} ticks per full iteration), epilogue {100 * (tot - r[3 + nt]) / tot:.1f} %")
