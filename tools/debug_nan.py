import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mkg_analogy_amd import ops, data_synth as D
from mkg_analogy_amd.trainer import Trainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
P = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda", 0)
model, lit, cfg = bench.build(P, seed=0, device=dev, backbone="mkgformer")
batch = D.make_batch(B, 64, seed=1234, device=dev)
NS = int(sys.argv[3]) if len(sys.argv) > 3 else 150
tr = Trainer(max_epochs=1, max_steps=NS, world_size=1)
tr._setup(lit, [None] * NS)
st = model.store
for i in range(int(sys.argv[4]) if len(sys.argv) > 4 else 14):
    loss = tr.train_step(lit, batch, i)
    torch.cuda.synchronize()
    g = st.grad
    bad = (~torch.isfinite(g)).sum().item()
    print(f"step {i}: loss {float(loss):.5f} lr {tr.optimizer.param_groups[0]['lr']:.3e} grad nonfinite {bad} |g| {g.float().norm().item():.4e} master nonfinite {(~torch.isfinite(st.master)).sum().item()}")
    if bad:
        for name, s in st.slots.items():
            gg = g[s.offset:s.offset + s.numel]
            nb = (~torch.isfinite(gg)).sum().item()
            if nb:
                print("   ", name, nb, "of", s.numel)
        break
