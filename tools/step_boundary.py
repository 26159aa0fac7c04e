"""Un-profiled timeline of the step boundary: where does the main queue stand between the last backward kernel of step N and
the first forward kernels of step N+1?  HIP events on the compute stream, no profiler attached (rocprofv3 slows the host enough to open gaps
that the un-profiled run does not have).

    bwd_end   recorded when loss.backward() has enqueued its last main-stream kernel
    step_end  after optimizer.step() (the main stream has joined the AdamW stream) + scheduler.step()
    fwd_k1    after the first forward kernel of the next step (patchify_k)
    fwd_emb   after the vision pre-LayerNorm (patch GEMM, assemble, LN: the first ~0.5 ms of forward work)

usage: python tools/step_boundary.py [steps] ; MART_ASYNC_STEP=1 for the off-stream zero-fill / W^T refresh"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mkg_analogy_amd import data_synth as D, ops  # noqa: E402
from mkg_analogy_amd.trainer import Trainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
model, lit, cfg = bench.build(16, seed=0, device=dev, entity_head=D.N_ENT)
batch = D.make_batch(256, 64, seed=1234, device=dev, n_labels=D.N_ENT)
tr = Trainer(max_epochs=1, max_steps=10 * (steps + 8))
tr._setup(lit, [None] * (10 * (steps + 8)))
ev = lambda: torch.cuda.Event(enable_timing=True)
marks = []
cur = {}
orig_patchify, orig_ln = ops.patchify, ops.ln_fwd
state = {"ln": 0}


def patchify(*a, **k):
    r = orig_patchify(*a, **k)
    e = ev(); e.record(); cur["fwd_k1"] = e
    state["ln"] = 0
    return r


def ln_fwd(**k):
    r = orig_ln(**k)
    if state["ln"] == 0 and "fwd_k1" in cur and "fwd_emb" not in cur:
        e = ev(); e.record(); cur["fwd_emb"] = e
    state["ln"] += 1
    return r


ops.patchify, ops.ln_fwd = patchify, ln_fwd
for i in range(steps + 4):
    cur = {}
    e0 = ev(); e0.record(); cur["start"] = e0
    lit.model.train()
    tr.optimizer.zero_grad()
    eng = lit.model.engine
    tr.sync.begin()
    tr.optimizer.begin_step()
    eng.grad_ready_async = tr.optimizer.ready
    loss = lit.training_step(dict(batch), i)
    loss.backward()
    e1 = ev(); e1.record(); cur["bwd_end"] = e1
    tr.sync.finish()
    tr.optimizer.step()
    tr.scheduler.step()
    e2 = ev(); e2.record(); cur["step_end"] = e2
    marks.append(cur)
torch.cuda.synchronize()
marks = marks[4:]
rows = []
for a, b in zip(marks[:-1], marks[1:]):
    rows.append((a["bwd_end"].elapsed_time(a["step_end"]), a["step_end"].elapsed_time(b["fwd_k1"]), b["fwd_k1"].elapsed_time(b["fwd_emb"]),
                 a["start"].elapsed_time(b["start"])))
n = len(rows)
avg = [sum(r[j] for r in rows) / n for j in range(4)]
print(f"MART_ASYNC_STEP={os.environ.get('MART_ASYNC_STEP', '0')}  {n} step boundaries, ms on the compute stream (mean / max):")
for j, name in enumerate(("bwd_end -> step_end   (exposed AdamW tail + joins; in-order: + W^T refresh)", "step_end -> fwd_k1    (in-order: zero-fill; + patchify_k ~0.12 ms)",
                          "fwd_k1 -> fwd_emb     (patch GEMM, assemble, pre-LN)", "step period")):
    print(f"   {name:78s} {avg[j]:8.3f} / {max(r[j] for r in rows):8.3f}")
