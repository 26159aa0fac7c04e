"""Per-step view of the caching allocator under the training step: hipMalloc count, reserved / active bytes, and the sizes the pool grew by.
Usage: python tools/alloc_trace.py [steps] [seq_len]  (run on a GPU box).  Evidence: profiles/r06_alloc_trace.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from mkg_analogy_amd import data_synth as D
from mkg_analogy_amd.trainer import Trainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
L = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
model, lit, cfg = B.build(int(os.environ.get("PATCH", "16")), seed=0, device=dev, backbone="mkgformer", entity_head=11292)
D.load_seeded_weights(model, lit, seed=0, conditioned=True)
batch = D.make_batch(256, L, seed=1234, device=dev, pretrain=False, n_labels=11292)
tr = Trainer(max_epochs=1, max_steps=10 * steps, world_size=1)
tr._setup(lit, [None] * (10 * steps))
prev = torch.cuda.memory_stats()
for i in range(steps):
    t0 = time.perf_counter()
    tr.train_step(lit, batch, i)
    if os.environ.get("SYNC_EACH", "0") == "1" or i < int(os.environ.get("SYNC_FIRST", "0")):
        torch.cuda.synchronize()                                   # SYNC_FIRST=3: a cold process, whose first steps run with the host not ahead
    st = torch.cuda.memory_stats()
    print(f"step {i:3d}  host {1e3 * (time.perf_counter() - t0):7.2f} ms  device_alloc +{st['num_device_alloc'] - prev['num_device_alloc']:3d}  free +{st['num_device_free'] - prev['num_device_free']:3d}  "
          f"reserved {st['reserved_bytes.all.current'] / 2**30:7.2f} GiB  active {st['active_bytes.all.current'] / 2**30:7.2f} GiB  "
          f"alloc_retries {st['num_alloc_retries']}  segments {st['segment.all.current']}", flush=True)
    prev = st
torch.cuda.synchronize()
print(f"live bytes at the end of a forward pass (Trainer._live_peak) {tr._live_peak / 2**30:.2f} GiB; peak of live bytes over the run {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")
print(torch.cuda.memory_summary(abbreviated=True))
