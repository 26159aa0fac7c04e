"""Where the fp32-accurate evaluation pass spends its time: three validation passes over the bench batch at precision fp32 (run under rocprofv3
--kernel-trace by tools/profile_eval_fp32.sh).  usage: python tools/eval_fp32_profile.py [passes]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mkg_analogy_amd import data_synth as D  # noqa: E402
from mkg_analogy_amd.trainer import Trainer  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda", 0)
model, lit, cfg = bench.build(16, seed=0, device=dev, entity_head=D.N_ENT)
batch = D.make_batch(256, 64, seed=1234, device=dev, n_labels=D.N_ENT)
tr = Trainer(max_epochs=1, max_steps=10)
tr._setup(lit, [None] * 10)
for prec in os.environ.get("EVAL_PRECS", "bf16,fp32").split(","):       # EVAL_PRECS=fp32: a trace of the fp32-accurate passes only
    lit.args.eval_precision = prec
    tr.validate(lit, [batch])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        m = tr.validate(lit, [batch])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{prec}: {256 / dt:.1f} examples/s ({1000 * dt:.2f} ms per pass) hits1 {m.get('Eval_entity/hits1')} mean_rank {m.get('Eval_entity/mean_rank')}", flush=True)
