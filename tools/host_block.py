"""Where does the HOST spend the forward pass of a training step?  Wraps every ops.* entry point, the engine's allocator helper and its stream helpers with
perf_counter timers (un-profiled run).  usage: python tools/host_block.py [--patch 32]"""
import argparse
import collections
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mkg_analogy_amd import data_synth as D, engine as E, ops  # noqa: E402
from mkg_analogy_amd.trainer import Trainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--patch", type=int, default=32)
ap.add_argument("--steps", type=int, default=6)
a = ap.parse_args()
dev = torch.device("cuda", 0)
model, lit, cfg = bench.build(a.patch, 0, dev, entity_head=D.N_ENT)
D.load_seeded_weights(model, lit, seed=0, conditioned=True)
batch = D.make_batch(256, 64, seed=1234, device=dev, n_labels=D.N_ENT)
tr = Trainer(max_epochs=1, max_steps=1000, world_size=1)
tr._setup(lit, [None] * 1000)
for i in range(4):
    tr.train_step(lit, batch, i)
torch.cuda.synchronize()
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])


def wrap(owner, name, tag):
    f = getattr(owner, name)

    def g(*x, **kw):
        t = time.perf_counter()
        try:
            return f(*x, **kw)
        finally:
            d = time.perf_counter() - t
            r = acc[tag]
            r[0] += 1; r[1] += d; r[2] = max(r[2], d)
    setattr(owner, name, g)


for n in dir(ops):
    f = getattr(ops, n)
    if callable(f) and not n.startswith("_") and getattr(f, "__module__", "") == ops.__name__:
        wrap(ops, n, "ops." + n)
wrap(E, "_e", "engine._e (torch.empty)")
eng = model.engine
for n in ("_pass_begin", "_pass_end", "_text_begin", "_text_done", "_text_record", "_main_record", "_text_wait", "_main_wait", "_join", "_tn"):
    wrap(eng, n, "engine." + n)
wrap(tr.optimizer, "zero_grad", "optimizer.zero_grad")
wrap(tr.optimizer, "step", "optimizer.step")
wrap(model.store, "join_pending", "store.join_pending")
ms0 = torch.cuda.memory_stats()
t0 = time.perf_counter()
for i in range(a.steps):
    tr.train_step(lit, batch, 4 + i)
t1 = time.perf_counter()
torch.cuda.synchronize()
ms1 = torch.cuda.memory_stats()
for k in ("num_alloc_retries", "num_ooms", "segment.all.allocated", "segment.all.freed", "num_sync_all_streams", "num_device_alloc", "num_device_free"):
    print(f"  {k}: {ms0.get(k)} -> {ms1.get(k)}")
print(f"  reserved {ms1['reserved_bytes.all.current'] / 2**30:.1f} GiB (peak {ms1['reserved_bytes.all.peak'] / 2**30:.1f}), allocated peak {ms1['allocated_bytes.all.peak'] / 2**30:.1f} GiB, "
      f"active {ms1['active_bytes.all.current'] / 2**30:.1f} GiB, inactive split {ms1['inactive_split_bytes.all.current'] / 2**30:.1f} GiB")
print(f"patch {a.patch}: host {1e3 * (t1 - t0) / a.steps:.2f} ms per step; top host-time entries per step (calls, ms total, max single ms):")
for k, (c, s, m) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:18]:
    print(f"  {k:34s} {c / a.steps:7.1f} {1e3 * s / a.steps:9.3f} {1e3 * m:9.3f}")
