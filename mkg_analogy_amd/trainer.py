"""A ~100-line stand-in for the pl.Trainer loop the reference relies on (MarT/main.py:151-162):
drives TransformerLitModel's hooks (training_step / validation_step / *_epoch_end / configure_optimizers),
steps the scheduler every batch (interval 'step'), averages gradients across ranks when launched under torchrun."""
from __future__ import annotations

import os
import time
from typing import Dict, Iterable, List, Optional

import torch

from . import functional as Fn
from .distributed import GradSync, all_gather_ranks


class Trainer:
    def __init__(self, max_epochs: int = 1, max_steps: Optional[int] = None, accumulate_grad_batches: int = 1,
                 world_size: Optional[int] = None, log_every: int = 0):
        self.max_epochs, self.max_steps, self.accumulate_grad_batches = max_epochs, max_steps, accumulate_grad_batches
        if world_size is None:                                      # one process per GPU under torchrun: the process group says how many
            import torch.distributed as dist
            world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.world_size = world_size
        self.log_every = log_every
        import os
        self.stream_optimizer = os.environ.get("MART_STREAM_OPT", "1") == "1"   # AdamW ranges launched under the backward pass
        self.num_train_batches = 0
        self.global_step = 0
        self._steps_seen = 0
        self._live_peak = 0                                            # live device bytes at the end of a forward pass (first three steps)
        self.history: List[Dict[str, float]] = []

    def _setup(self, lit, train_batches):
        lit.trainer = self
        # num_training_steps (lit_models/base.py:74-95) divides the UNSHARDED loader length by the device count: a loader that is
        # already sharded by a DistributedSampler (data_module.KGC._loader) reports its per-rank length
        self.num_train_batches = len(train_batches) * self._shards(train_batches)
        lit.model.finalize()
        oc = lit.configure_optimizers()
        self.optimizer, self.scheduler = oc["optimizer"], oc["lr_scheduler"]["scheduler"]
        self.sync = GradSync(lit.model)
        self.sync.broadcast_optimizer(self.optimizer)
        self.optimizer.grad_scale = self.sync.grad_scale / max(1, self.accumulate_grad_batches)
        if hasattr(self.optimizer, "fused_zero_grad"):
            # this loop owns zero_grad() and step(): the update zeroes the gradients it consumes and the next zero_grad() skips its fill (optim.FusedAdamW)
            import os
            self.optimizer.fused_zero_grad = os.environ.get("MART_FUSED_ZERO_GRAD", "1") == "1"

    @staticmethod
    def _shards(loader) -> int:
        sampler = getattr(loader, "sampler", None)
        return int(getattr(sampler, "num_replicas", 1) or 1) if sampler is not None and hasattr(sampler, "set_epoch") else 1

    def train_step(self, lit, batch, batch_idx: int, end_of_epoch: bool = False) -> torch.Tensor:
        lit.model.train()
        first = batch_idx % self.accumulate_grad_batches == 0
        last = (batch_idx + 1) % self.accumulate_grad_batches == 0 or end_of_epoch    # PL steps on an epoch's last (partial) window too
        if first:
            self.optimizer.zero_grad()
        eng = lit.model.engine
        stream_opt = self.stream_optimizer and hasattr(self.optimizer, "begin_step")
        if last:
            self.sync.begin()
            if self.sync.reducer is not None:
                eng.grad_ready_async = self.sync.reducer.ready
            if stream_opt:
                # parameter ranges are updated as soon as their gradients are final (and, under DDP, all-reduced), below the
                # rest of the backward pass
                self.optimizer.begin_step()
                if self.sync.reducer is not None:
                    self.sync.reducer.on_bucket = self.optimizer.ready
                else:
                    eng.grad_ready_async = self.optimizer.ready
            elif self.sync.reducer is not None:
                self.sync.reducer.on_bucket = None
        else:
            eng.grad_ready = None
            eng.grad_ready_async = None
        loss = lit.training_step(dict(batch), batch_idx)
        if self._steps_seen < 3 and loss.is_cuda:
            self._live_peak = max(self._live_peak, torch.cuda.memory_allocated())   # everything the backward pass needs is alive here
        loss.backward()
        if last:
            self.sync.finish()
            self.optimizer.step()
            self.scheduler.step()
            self.global_step += 1
            if self.sync.reducer is not None:
                eng.grad_ready_async = self.sync.reducer.ready
        self._steps_seen += 1
        if self._steps_seen == 3:
            self._pool_headroom()
        return loss.detach()

    def settle_pool(self) -> None:
        """For a caller that is about to time steps after fewer than three warm-up steps: apply the pool headroom now instead of inside its timed region."""
        if self._steps_seen < 3:
            self._pool_headroom()
            self._steps_seen = 3

    def _pool_headroom(self) -> None:
        """Once, after the third step: bring the caching allocator's pool to what two steps in flight need -- MART_POOL_FACTOR (default 2.5) x the live
        bytes at the end of a forward pass (this trainer's own steps; timing-independent) -- with one allocation per stream pool that is handed
        straight back.  A block that a side queue touched returns to the pool only when that queue has passed the free (record_stream), so the pool
        of a host that runs ahead holds two steps' worth of buffers plus whatever size classes happened to be pending when they were asked for:
        without this it grows by a hipMalloc about once per step for 25 steps (78.3 -> 82.4 GiB at B = 256), and when the first steps ran with the
        host NOT ahead (a cold process: libraries paging in) the whole second step's worth (32 GiB, ~140 allocations) is allocated later, in
        whatever steps are being timed (tools/alloc_trace.py, profiles/r06_alloc_trace.txt).  With the slabs cached those requests are carved from
        them.  Nothing happens when the pool is already that large (a process that has run something bigger before).  MART_POOL_HEADROOM=0: off."""
        if os.environ.get("MART_POOL_HEADROOM", "1") in ("0", "0.0") or not torch.cuda.is_available() or self._live_peak <= 0:
            return
        # the allocator keeps one pool per stream (a block serves only the stream it was allocated on): the deficit is split over the streams in
        # proportion to what each has reserved so far, and each share is allocated -- and freed -- on its own stream
        dev = torch.cuda.current_device()
        by_stream = {}
        for seg in torch.cuda.memory_snapshot():
            if seg.get("device", dev) == dev:
                by_stream[seg["stream"]] = by_stream.get(seg["stream"], 0) + seg["total_size"]
        reserved = sum(by_stream.values())
        if reserved <= 0:
            return
        deficit = float(os.environ.get("MART_POOL_FACTOR", "2.5")) * self._live_peak - reserved
        free, _ = torch.cuda.mem_get_info()
        if deficit < (64 << 20) or deficit > free // 2:
            return
        for ptr, r in by_stream.items():
            want = int(deficit * r / reserved)
            if want < (32 << 20):
                continue
            stream = torch.cuda.ExternalStream(ptr) if ptr else torch.cuda.default_stream()
            try:
                with torch.cuda.stream(stream):
                    slab = torch.empty(want, dtype=torch.uint8, device="cuda")
                    del slab
            except torch.cuda.OutOfMemoryError:
                return

    def fit(self, lit, train_batches: Iterable, val_batches: Optional[Iterable] = None):
        train_batches = list(train_batches) if not hasattr(train_batches, "__len__") else train_batches
        self._setup(lit, train_batches)
        for epoch in range(self.max_epochs):
            t0 = time.time()
            sampler = getattr(train_batches, "sampler", None)
            if hasattr(sampler, "set_epoch"):
                sampler.set_epoch(epoch)                              # a different shuffle per epoch, the same on every rank
            n = len(train_batches)
            for i, batch in enumerate(train_batches):
                loss = self.train_step(lit, batch, i, end_of_epoch=(i == n - 1))
                if self.log_every and (i + 1) % self.log_every == 0:
                    Fn.check_status()
                    print(f"epoch {epoch} step {i + 1}: loss {float(loss):.4f} lr {self.optimizer.param_groups[0]['lr']:.3e}")
                if self.max_steps and self.global_step >= self.max_steps:
                    break
            Fn.check_status()                                      # bad labels / examples without [MASK] seen during the epoch (one host sync)
            rec = {"epoch": epoch, "train_time_s": time.time() - t0}
            if val_batches is not None:
                rec.update(self.validate(lit, val_batches))
            self.history.append(rec)
            if self.max_steps and self.global_step >= self.max_steps:
                break
        return self.history

    def _run_eval(self, lit, batches, step_fn, end_fn) -> Dict[str, float]:
        lit.model.eval()
        outs = [step_fn(dict(b), i) for i, b in enumerate(batches)]
        Fn.check_status()                                          # (the ranks were copied to the host already: no extra sync)
        merged = {}
        sharded = self._shards(batches) > 1 or getattr(batches, "sharded", False)   # every rank saw the whole set otherwise: no gather
        for key in ("entity_ranks", "relation_ranks"):
            parts = [o[key] for o in outs if key in o]
            if parts:
                import numpy as np
                merged[key] = all_gather_ranks(np.concatenate(parts)) if sharded else np.concatenate(parts)
        lit.logged = {}
        end_fn([merged])
        return dict(lit.logged)

    def validate(self, lit, batches) -> Dict[str, float]:
        return self._run_eval(lit, batches, lit.validation_step, lit.validation_epoch_end)

    def test(self, lit, batches) -> Dict[str, float]:
        return self._run_eval(lit, batches, lit.test_step, lit.test_epoch_end)
