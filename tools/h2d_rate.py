"""PCIe-inclusive step rate of the fine-tune workload when the batch arrives as host buffers (the reference's
pixel_values [B,2,3,224,224] fp32 batches, 308 MB at B=256): pinned and pageable H2D rates, the step with the copy in line,
and the step with the copy of batch i+1 on a copy stream under step i."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
B = 256
pix = torch.randn(B, 2, 3, 224, 224)
pin = pix.pin_memory()
dst = torch.empty_like(pix, device=dev)
for name, src in (("pageable", pix), ("pinned", pin)):
    for _ in range(2): dst.copy_(src, non_blocking=True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"H2D {name}: {pix.numel() * 4 / dt / 1e9:.1f} GB/s, {dt * 1e3:.2f} ms per 308 MB batch")

from mkg_analogy_amd import data_synth as D
from mkg_analogy_amd.trainer import Trainer
model, lit, cfg = bench.build(16, seed=0, device=dev, backbone="mkgformer", entity_head=D.N_ENT)
batch = D.make_batch(B, 64, seed=1234, device=dev, n_labels=D.N_ENT)
tr = Trainer(max_epochs=1, max_steps=1000, world_size=1)
tr._setup(lit, [None] * 1000)
steps = 8
def run(mode):
    for i in range(2): tr.train_step(lit, batch, i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    cs = torch.cuda.Stream()
    nxt = torch.empty_like(dst)
    for i in range(steps):
        if mode == "inline":
            batch["pixel_values"].copy_(pin, non_blocking=True)
        elif mode == "prefetch":
            with torch.cuda.stream(cs):
                nxt.copy_(pin, non_blocking=True)
        tr.train_step(lit, batch, 2 + i)
        if mode == "prefetch":
            torch.cuda.current_stream().wait_stream(cs)
            batch["pixel_values"], nxt = nxt, batch["pixel_values"]
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for mode in ("resident", "inline", "prefetch", "resident"):
    print(f"step, batch {mode}: {run(mode):.2f} ms")
