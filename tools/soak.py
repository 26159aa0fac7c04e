"""Training soak at the bench shape: N optimizer steps on ONE fixed batch (the timed default configuration), loss every 25 steps; every value finite and the
loss falling (the network memorises the batch).  usage: python tools/soak.py [steps]"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mkg_analogy_amd import data_synth as D  # noqa: E402
from mkg_analogy_amd.trainer import Trainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
model, lit, cfg = bench.build(16, seed=0, device=dev, entity_head=D.N_ENT)
lit.args.lr = 5e-5
batch = D.make_batch(256, 64, seed=1234, device=dev, n_labels=D.N_ENT)
tr = Trainer(max_epochs=1, max_steps=steps)
tr._setup(lit, [None] * steps)
losses = []
for i in range(steps):
    loss = tr.train_step(lit, dict(batch), i)
    if i % 25 == 0 or i == steps - 1:
        losses.append(float(loss))
        print(f"step {i:4d} loss {losses[-1]:.4f}", flush=True)
torch.cuda.synchronize()
st = model.store
assert all(math.isfinite(x) for x in losses), losses
assert bool(torch.isfinite(st.master).all()) and bool(torch.isfinite(st.shadow.float()).all()) and bool(torch.isfinite(st.shadow_h.float()).all())
print("fp16 shadow |max|", float(st.shadow_h.float().abs().max()), " bf16 shadow |max|", float(st.shadow.float().abs().max()))
assert losses[-1] < 0.7 * losses[0], (losses[0], losses[-1])
print("SOAK OK")
