"""Un-profiled timeline marks of the training step (HIP events recorded on the main queue at phase boundaries of engine.forward / engine.backward):
where the main queue spends the step, without rocprofv3's per-launch host cost.  usage: python tools/tail_marks.py [--patch 32] [--steps 10]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mkg_analogy_amd import data_synth as D  # noqa: E402
from mkg_analogy_amd.trainer import Trainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--patch", type=int, default=32)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--seq-len", type=int, default=64)
a = ap.parse_args()
dev = torch.device("cuda", 0)
model, lit, cfg = bench.build(a.patch, 0, dev, entity_head=D.N_ENT)
D.load_seeded_weights(model, lit, seed=0, conditioned=True)
batch = D.make_batch(a.batch, a.seq_len, seed=1234, device=dev, n_labels=D.N_ENT)
tr = Trainer(max_epochs=1, max_steps=1000, world_size=1)
tr._setup(lit, [None] * 1000)
for i in range(5):
    tr.train_step(lit, batch, i)
torch.cuda.synchronize()
eng = model.engine
eng.marks = []
import time
t0 = time.perf_counter()
for i in range(a.steps):
    tr.train_step(lit, batch, 5 + i)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / a.steps * 1e3
marks, eng.marks = eng.marks, None
names = [m[0] for m in marks]
per = len(marks) // a.steps
acc, host = {}, {}
for s in range(1, a.steps):                         # skip the first marked step
    seg = marks[s * per:(s + 1) * per]
    nxt = marks[(s + 1) * per][1] if (s + 1) * per < len(marks) else None
    for (n0, e0, h0), (n1, e1, h1) in zip(seg[:-1], seg[1:]):
        acc.setdefault(f"{n0} -> {n1}", []).append(e0.elapsed_time(e1))
        host.setdefault(f"{n0} -> {n1}", []).append((h1 - h0) * 1e3)
    if nxt is not None:
        acc.setdefault(f"{seg[-1][0]} -> next fwd_begin", []).append(seg[-1][1].elapsed_time(nxt))
        host.setdefault(f"{seg[-1][0]} -> next fwd_begin", []).append((marks[(s + 1) * per][2] - seg[-1][2]) * 1e3)
print(f"patch {a.patch}: {wall:.3f} ms per step (wall, {a.steps} steps); main-queue intervals, mean over {a.steps - 1} steps [ms]:")
tot = 0.0
for k, v in acc.items():
    m = sum(v) / len(v)
    tot += m
    hv = host.get(k, [0.0])
    print(f"  {k:42s} gpu {m:8.3f}   (min {min(v):.3f} max {max(v):.3f})   host enqueue {sum(hv) / len(hv):8.3f}")
print(f"  sum {tot:.3f}")
