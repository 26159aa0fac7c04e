"""Per-shape in-step timing of gemm_nt / gemm_tn (HIP events around every launch of one fine-tune step)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mkg_analogy_amd import ops, data_synth as D
from mkg_analogy_amd.trainer import Trainer
ops.require_gpu()
dev = torch.device("cuda", 0)
model, lit, cfg = bench.build(int(os.environ.get("PATCH", 16)), seed=0, device=dev, backbone="mkgformer")
batch = D.make_batch(256, 64, seed=1234, device=dev)
tr = Trainer(max_epochs=1, max_steps=100, world_size=1)
tr._setup(lit, [None] * 100)
for i in range(2):
    tr.train_step(lit, batch, i)
rec = []
orig_nt, orig_tn = ops.gemm_nt, ops.gemm_tn
def nt(A, B, out, **kw):
    K = A.shape[-1]; K2 = kw["A2"].shape[-1] if kw.get("A2") is not None else 0
    M = kw.get("M") or (kw["a_rows"].numel() if kw.get("a_rows") is not None else A.shape[-2])
    N = kw.get("N") or (kw["b_rows"].numel() if kw.get("b_rows") is not None else B.shape[-2])
    bz = kw.get("batch", 1)
    epi = "+".join(k for k in ("bias", "preact", "mulz", "res_f32", "res_bf16", "C2") if kw.get(k) is not None)
    if kw.get("act", 0): epi += "+act"
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); r = orig_nt(A, B, out, **kw); e.record()
    rec.append((("NT", M, N, K + K2, bz, str(out.dtype)[6:], epi), 2.0 * M * N * (K + K2) * bz, s, e))
    return r
def tn(X, Y, out, **kw):
    M = kw.get("M") or X.shape[-2]; NX = kw.get("NX") or X.shape[-1]; NY = kw.get("NY") or Y.shape[-1]; bz = kw.get("batch", 1)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); r = orig_tn(X, Y, out, **kw); e.record()
    rec.append((("TN", M, NX, NY, bz, "f32", "colsum" if kw.get("colsum") is not None else ""), 2.0 * M * NX * NY * bz, s, e))
    return r
ops.gemm_nt, ops.gemm_tn = nt, tn
import mkg_analogy_amd.engine as E, mkg_analogy_amd.functional as F
tr.train_step(lit, batch, 2)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, fl, s, e in rec:
    a = agg.setdefault(key, [0, 0.0, 0.0]); a[0] += 1; a[1] += s.elapsed_time(e); a[2] += fl
tot = sum(a[1] for a in agg.values())
print(f"{'kind M N K batch out epilogue':70s} calls  total_ms  avg_ms   TF/s   %")
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{str(key):70s} {a[0]:5d} {a[1]:9.2f} {a[1]/a[0]:7.3f} {a[2]/a[1]/1e9:7.0f} {100*a[1]/tot:5.1f}")
print(f"total GEMM ms per step: {tot:.1f}")
