import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mkg_analogy_amd import ops, data_synth as D
from mkg_analogy_amd.trainer import Trainer
NS = int(sys.argv[1])
dev = torch.device("cuda", 0)
model, lit, cfg = bench.build(16, seed=0, device=dev, backbone="mkgformer")
batch = D.make_batch(256, 64, seed=1234, device=dev)
tr = Trainer(max_epochs=1, max_steps=NS, world_size=1)
tr._setup(lit, [None] * NS)
st = model.store
def fin(name, t): print(f"   {name}: nonfinite {(~torch.isfinite(t.float())).sum().item()} absmax {t.float().abs().max().item():.4e}")
print("lr before first update", tr.optimizer.param_groups[0]["lr"])
loss = tr.train_step(lit, batch, 0); torch.cuda.synchronize()
print("step 0 loss", float(loss), "lr now", tr.optimizer.param_groups[0]["lr"])
fin("master", st.master); fin("shadow", st.shadow); fin("shadow_t", st.shadow_t); fin("m", tr.optimizer.m); fin("v", tr.optimizer.v); fin("grad", st.grad)
model.eval()
with torch.no_grad():
    l2 = lit.training_step(dict(batch), 1)
print("eval-mode loss after update 0:", float(l2))
model.train()
loss = tr.train_step(lit, batch, 1); torch.cuda.synchronize()
print("step 1 loss", float(loss))
fin("master", st.master); fin("grad", st.grad)
