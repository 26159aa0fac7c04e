"""Training steps over batches whose padded length changes from batch to batch, as the reference's collate makes them (data_module.py:113-119 pads a
batch to its longest example: 40 .. 57 tokens on MARS), against the same number of steps at the fixed maximum.  Run on a GPU box.
Evidence: profiles/r06_var_len.txt"""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from mkg_analogy_amd import data_synth as D
from mkg_analogy_amd import functional as Fn
from mkg_analogy_amd.trainer import Trainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
patch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda:0")
model, lit, cfg = B.build(patch, seed=0, device=dev, backbone="mkgformer", entity_head=11292)
D.load_seeded_weights(model, lit, seed=0, conditioned=True)
lens = list(range(40, 58))
batches = {L: D.make_batch(256, L, seed=1234 + L, device=dev, pretrain=False, n_labels=11292) for L in lens}
tr = Trainer(max_epochs=1, max_steps=100 * steps, world_size=1)
tr._setup(lit, [None] * (100 * steps))
rng = random.Random(0)


def run(seq, tag, first):
    torch.cuda.synchronize()
    a0 = torch.cuda.memory_stats()["num_device_alloc"]
    t0 = time.perf_counter()
    for i, L in enumerate(seq):
        loss = tr.train_step(lit, batches[L], first + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = torch.cuda.memory_stats()
    print(f"{tag:34s} {len(seq)} steps  {1e3 * dt / len(seq):7.2f} ms/step  {256 * len(seq) / dt:7.0f} examples/s  tokens/step {sum(seq) / len(seq):5.1f}  "
          f"hipMalloc +{st['num_device_alloc'] - a0}  reserved {st['reserved_bytes.all.current'] / 2**30:.1f} GiB  loss {float(loss):.4f}", flush=True)


n = 0
run([57] * 6, "warm-up, L = 57", n); n += 6
run([57] * steps, "fixed L = 57", n); n += steps
seq = [rng.choice(lens) for _ in range(steps)]
run(seq, "L drawn from 40..57 per batch (1st)", n); n += steps
seq = [rng.choice(lens) for _ in range(steps)]
run(seq, "L drawn from 40..57 per batch (2nd)", n); n += steps
run([57] * steps, "fixed L = 57 again", n); n += steps
Fn.check_status()
