"""cProfile of the host side of one training step (the GPU runs asynchronously; this is pure enqueue cost)."""
import sys, os, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mkg_analogy_amd import ops, data_synth as D
from mkg_analogy_amd.trainer import Trainer
ops.require_gpu()
dev = torch.device("cuda", 0)
B = int(os.environ.get("B", 256))
model, lit, cfg = bench.build(int(os.environ.get("PATCH", 16)), seed=0, device=dev, backbone="mkgformer")     # PATCH=32: the 49-patch geometry
batch = D.make_batch(B, 64, seed=1234, device=dev)
tr = Trainer(max_epochs=1, max_steps=1000, world_size=1)
tr._setup(lit, [None] * 1000)
for i in range(3):
    tr.train_step(lit, batch, i)
torch.cuda.synchronize()
model.engine.max_inflight = 0
t0 = time.perf_counter()
for i in range(3):
    tr.train_step(lit, batch, 3 + i)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host enqueue per step (no profiler): {1e3 * (t1 - t0) / 3:.1f} ms")
pr = cProfile.Profile()
pr.enable()
for i in range(3):
    tr.train_step(lit, batch, 6 + i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
