"""Per-step wall time and allocator state over a longer run of the bench step (looks for drift)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mkg_analogy_amd import ops, data_synth as D
from mkg_analogy_amd.trainer import Trainer
ops.require_gpu()
dev = torch.device("cuda", 0)
model, lit, cfg = bench.build(16, seed=0, device=dev, backbone="mkgformer")
batch = D.make_batch(256, 64, seed=1234, device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
tr = Trainer(max_epochs=1, max_steps=10 * n, world_size=1)
tr._setup(lit, [None] * (10 * n))
for i in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss = tr.train_step(lit, batch, i)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = torch.cuda.memory_stats()
    print(f"step {i:3d} {dt*1e3:8.1f} ms  reserved {torch.cuda.memory_reserved()/2**30:7.1f} GiB allocated {torch.cuda.memory_allocated()/2**30:7.1f} GiB  "
          f"mallocs {st.get('num_device_alloc', 0)} frees {st.get('num_device_free', 0)} retries {st.get('num_alloc_retries', 0)} loss {float(loss):.4f}", flush=True)

# ---- same loop without the per-step synchronisation (what bench.py times): host runs ahead of the GPU
for chunk in [int(c) for c in os.environ.get("CHUNKS", "10,20,40").split(",")]:
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(chunk):
        loss = tr.train_step(lit, batch, n + i)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    st = torch.cuda.memory_stats()
    print(f"unsynced x{chunk}: host enqueue {1e3*(t1-t0)/chunk:7.1f} ms/step, total {1e3*(t2-t0)/chunk:7.1f} ms/step  reserved {torch.cuda.memory_reserved()/2**30:7.1f} GiB "
          f"mallocs {st.get('num_device_alloc', 0)} frees {st.get('num_device_free', 0)} retries {st.get('num_alloc_retries', 0)}", flush=True)
