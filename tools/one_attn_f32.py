"""The f32 matrix-pipe attention kernel alone at the bench's vision shape (for tools/pmc.sh / timing): B=256, 12 heads, 393 queries x (64 prefix + 393) keys."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mkg_analogy_amd import ops
ops.require_gpu()
B, nh, H, Nv, L = 256, 12, 768, 393, 64
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B * Nv, 3 * H, device="cuda", generator=g)
pre = torch.randn(B * L, 3 * H, device="cuda", generator=g)
ctx = torch.empty(B * Nv, H, device="cuda")
kw = dict(q=qkv[:, :H], k=qkv[:, H:2 * H], v=qkv[:, 2 * H:], ctx=ctx, B=B, nh=nh, D=64, Sq=Nv, Sk=Nv, scale=0.125, pk=pre[:, H:2 * H], pv=pre[:, 2 * H:], Lp=L)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
kw["fast"] = len(sys.argv) > 2 and sys.argv[2] == "fast"      # two-term bf16 splits (evaluation passes) instead of the exact f32 kernel
ops.attn_fwd_f32(**kw)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n):
    ops.attn_fwd_f32(**kw)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
fl = 4.0 * B * nh * Nv * (Nv + L) * 64
print(f"attn_f32{' fast' if kw['fast'] else ''} (393 x 457): {1000 * dt:.3f} ms, {fl / dt / 1e12:.1f} TF/s of 157")
