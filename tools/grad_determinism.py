"""Which gradient tensors differ between two backward passes from the same weights on the same batch (eval mode)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mkg_analogy_amd import data_synth as D
dev = torch.device("cuda", 0)
patch = int(os.environ.get("PATCH", 32)); B = int(os.environ.get("B", 16))
model, lit, cfg = bench.build(patch, seed=0, device=dev)
model.eval()
gb = D.make_batch(B, 64, seed=51, device=dev)
st = model.store
grads = []
for _ in range(3):
    st.zero_grad()
    loss = lit.training_step(dict(gb), 1)
    loss.backward()
    torch.cuda.synchronize()
    grads.append((float(loss), st.grad.clone()))
print("losses", [g[0] for g in grads])
for r in (1, 2):
    names = []
    for n, sl in st.slots.items():
        a, b = grads[0][1][sl.offset:sl.offset + sl.numel], grads[r][1][sl.offset:sl.offset + sl.numel]
        if not torch.equal(a, b):
            names.append((n, int((a != b).sum()), sl.numel))
    print(f"run {r}: {len(names)} tensors differ")
    for n in names[:60]:
        print("   ", n)
