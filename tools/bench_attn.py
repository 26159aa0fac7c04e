"""Vision attention forward / backward at the bench shape (B=256, 12 heads, 393 queries, 393 or 64+393 keys), per-kernel times
through HIP events around the whole call (A/B two builds or MART_ATTN_RES=0/1 in two processes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mkg_analogy_amd import ops
from tools.bench_kernels import timeit

ops.require_gpu()
DEV, BF = "cuda", torch.bfloat16
B = int(os.environ.get("B", 256))
print("B =", B)
for (S, Lp) in ([(int(os.environ["S"]), 0), (int(os.environ["S"]), 64)] if "S" in os.environ else [(393, 0), (393, 64)]):
    nh, H = 12, 768
    bufs = []
    for i in range(4):                                    # rotate operands: the step never re-reads a hot buffer
        qkv = torch.randn(B * S, 3 * H, device=DEV).to(BF)
        tq = torch.randn(B * 64, 3 * H, device=DEV).to(BF)
        dctx = torch.randn(B * S, H, device=DEV).to(BF)
        bufs.append((qkv, tq, dctx))
    ctx = torch.empty(B * S, H, device=DEV, dtype=BF)
    lse = torch.empty(B, nh, S, device=DEV)
    dqkv = torch.empty(B * S, 3 * H, device=DEV, dtype=BF)
    dt = torch.empty(B * 64, 3 * H, device=DEV, dtype=BF)
    delta = torch.empty(B, nh, S, device=DEV)
    it = [0]

    def kw():
        qkv, tq, dctx = bufs[it[0] % 4]
        it[0] += 1
        return dict(q=qkv[:, :H], k=qkv[:, H:2*H], v=qkv[:, 2*H:], ctx=ctx, lse=lse, B=B, nh=nh, Sq=S, Sk=S, scale=0.125,
                    pk=tq[:, H:2*H] if Lp else None, pv=tq[:, 2*H:] if Lp else None, Lp=Lp), dctx
    ms = timeit(lambda: ops.attn_fwd(**kw()[0]))
    fl = 4 * B * nh * S * (S + Lp) * 64
    print(f"attn_fwd S={S} Lp={Lp}: {ms:.4f} ms {fl/ms/1e9:.0f} TF/s")

    def bwd():
        k, dctx = kw()
        ops.attn_bwd(dctx=dctx, delta=delta, dq=dqkv[:, :H], dk=dqkv[:, H:2*H], dv=dqkv[:, 2*H:],
                     dpk=dt[:, H:2*H] if Lp else None, dpv=dt[:, 2*H:] if Lp else None, **k)
    ops.attn_fwd(**kw()[0])
    ms = timeit(bwd)
    print(f"attn_bwd S={S} Lp={Lp}: {ms:.4f} ms {2.0*fl/ms/1e9:.0f} TF/s (algorithmic 2x fwd)")
