"""Micro-benchmarks of the hot kernels at BASELINE config-2 shapes (B=256, Nv=393, L=64).  GPU only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mkg_analogy_amd import ops

ops.require_gpu()
DEV = "cuda"
BF = torch.bfloat16


def timeit(fn, iters=20, warm=3):
    # the first config timed after an allocation phase used to read 10-15 % slow (clocks still ramping): warm for >= 50 ms
    t0 = time.perf_counter()
    n = 0
    while n < warm or time.perf_counter() - t0 < 0.05:
        fn()
        n += 1
        if n % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    B = int(os.environ.get("B", 256))
    Mv, Mt = B * 393, B * 64
    print(f"B={B} Mv={Mv} Mt={Mt}")
    for (M, N, K) in [(Mv, 2304, 768), (Mv, 768, 768), (Mv, 3072, 768), (Mv, 768, 3072), (Mt, 2304, 768), (Mt, 768, 768), (Mt, 3072, 768), (Mt, 768, 3072)]:
        A = torch.randn(M, K, device=DEV).to(BF)
        W = (torch.randn(N, K, device=DEV) * 0.02).to(BF)
        out = torch.empty(M, N, device=DEV, dtype=BF)
        bias = torch.randn(N, device=DEV)
        for cfg in (128, 256):
            ms = timeit(lambda: ops.gemm_nt(A, W, out, bias=bias, tile_cfg=cfg))
            print(f"gemm_nt M={M} N={N} K={K} cfg={cfg}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.0f} TF/s")
        ref = timeit(lambda: torch.nn.functional.linear(A, W))
        print(f"   torch(hipBLASLt) linear: {ref:.3f} ms {2*M*N*K/ref/1e9:.0f} TF/s")
    for (M, NX, NY) in [(Mv, 2304, 768), (Mv, 768, 768), (Mv, 3072, 768), (Mv, 768, 3072), (Mt, 3072, 768)]:
        X = (torch.randn(M, NX, device=DEV) * 0.1).to(BF)
        Y = torch.randn(M, NY, device=DEV).to(BF)
        out = torch.zeros(NX, NY, device=DEV)
        cs = torch.zeros(NX, device=DEV)
        ms = timeit(lambda: ops.gemm_tn(X, Y, out, colsum=cs))
        print(f"gemm_tn M={M} NX={NX} NY={NY}: {ms:.3f} ms  {2*M*NX*NY/ms/1e9:.0f} TF/s")
    for (S, Lp) in [(393, 0), (393, 64)]:
        nh, H = 12, 768
        qkv = torch.randn(B * S, 3 * H, device=DEV).to(BF)
        tq = torch.randn(B * 64, 3 * H, device=DEV).to(BF)
        ctx = torch.empty(B * S, H, device=DEV, dtype=BF)
        lse = torch.empty(B, nh, S, device=DEV)
        kw = dict(q=qkv[:, :H], k=qkv[:, H:2*H], v=qkv[:, 2*H:], ctx=ctx, lse=lse, B=B, nh=nh, Sq=S, Sk=S, scale=0.125,
                  pk=tq[:, H:2*H] if Lp else None, pv=tq[:, 2*H:] if Lp else None, Lp=Lp)
        ms = timeit(lambda: ops.attn_fwd(**kw))
        fl = 4 * B * nh * S * (S + Lp) * 64
        print(f"attn_fwd S={S} Lp={Lp}: {ms:.3f} ms {fl/ms/1e9:.0f} TF/s")
        dctx = torch.randn(B * S, H, device=DEV).to(BF)
        dqkv = torch.empty(B * S, 3 * H, device=DEV, dtype=BF)
        dt = torch.empty(B * 64, 3 * H, device=DEV, dtype=BF)
        delta = torch.empty(B, nh, S, device=DEV)
        ms = timeit(lambda: ops.attn_bwd(dctx=dctx, delta=delta, dq=dqkv[:, :H], dk=dqkv[:, H:2*H], dv=dqkv[:, 2*H:],
                                         dpk=dt[:, H:2*H] if Lp else None, dpv=dt[:, 2*H:] if Lp else None, **kw))
        print(f"attn_bwd S={S} Lp={Lp}: {ms:.3f} ms {2.5*fl/ms/1e9:.0f} TF/s (alg 2.5x fwd)")
    M, H = Mv, 768
    x = torch.randn(M, H, device=DEV)
    g, b = torch.ones(H, device=DEV), torch.zeros(H, device=DEV)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    ob = torch.empty(M, H, device=DEV, dtype=BF)
    ms = timeit(lambda: ops.ln_fwd(x_f32=x, gamma=g, beta=b, eps=1e-5, M=M, H=H, mean=mean, rstd=rstd, out_bf16=ob))
    print(f"ln_fwd M={M}: {ms:.3f} ms {M*H*6/ms/1e6:.0f} GB/s")
    dy = torch.randn(M, H, device=DEV).to(BF)
    ds, dsb = torch.empty(M, H, device=DEV), torch.empty(M, H, device=DEV, dtype=BF)
    dg, db = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
    ms = timeit(lambda: ops.ln_bwd(dy_bf16=dy, s=x, mean=mean, rstd=rstd, gamma=g, M=M, H=H, add_f32=x, ds_f32=ds, ds_bf16=dsb, dgamma=dg, dbeta=db))
    print(f"ln_bwd M={M}: {ms:.3f} ms {M*H*16/ms/1e6:.0f} GB/s")


if __name__ == "__main__":
    main()
