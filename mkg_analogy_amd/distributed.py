"""Data-parallel gradient synchronisation: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The reference has no collective code at all; under PL-DDP it would average gradients across ranks.
Here the flat fp32 gradient buffer is all-reduced in a few large contiguous buckets.  Buckets are issued on a side
stream as soon as the manual backward reports that every gradient below a flat offset is final (layout order ==
backward completion order, params.layout_order), so the collectives overlap the remaining backward kernels.
The 1/world_size factor is folded into the fused AdamW kernel (grad_scale), so no extra pass touches the buffer.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


class BucketedAllReduce:
    """All-reduce(sum) of ``flat[s:e]`` slices, launched in order as ``ready(upto)`` advances.

    ``bucket_dtype="bf16"`` (env MART_GRAD_BUCKET_DTYPE=bf16): every bucket is cast to a bf16 staging buffer, reduced, and cast
    back -- half the bytes per xGMI link (the fallback if RCCL picks a ring, which is per-link bound); fp32 is the
    default.  ``timing=True`` (MART_COMM_TIMING=1, bench.py for N > 1): start / end events per bucket on the collective stream and
    an end-of-backward event on the producer's stream give ``stats()``: the total time inside the collectives and the part of it
    that was EXPOSED (still running after the backward pass had finished: what the step actually waited for)."""

    def __init__(self, flat: torch.Tensor, buckets: List[Tuple[int, int]], group=None, bucket_dtype: Optional[str] = None,
                 timing: Optional[bool] = None):
        import os
        self.flat, self.buckets, self.group = flat, buckets, group
        self.next = 0
        self.handles = []
        self.side: Optional[torch.cuda.Stream] = torch.cuda.Stream() if flat.is_cuda else None
        self.on_bucket = None          # optional callback(end_offset): runs right after a bucket's all-reduce, on the stream that waits for it
        self.bucket_dtype = (bucket_dtype or os.environ.get("MART_GRAD_BUCKET_DTYPE", "fp32")).lower()
        assert self.bucket_dtype in ("fp32", "bf16"), "MART_GRAD_BUCKET_DTYPE: fp32 or bf16"
        self.timing = (os.environ.get("MART_COMM_TIMING", "0") == "1") if timing is None else bool(timing)
        self._stage = None
        self._ev: List[tuple] = []
        self._bwd_end = None
        self._stats = dict(steps=0, comm_ms=0.0, exposed_ms=0.0)

    def begin(self) -> None:
        self.next = 0
        self.handles = []
        self._ev = []

    def _reduce(self, s: int, e: int):
        """One bucket on the current stream: the collective (and the casts around it in bf16 mode)."""
        seg = self.flat[s:e]
        if self.bucket_dtype == "bf16":
            if self._stage is None:
                self._stage = torch.empty(max(b - a for a, b in self.buckets), device=self.flat.device, dtype=torch.bfloat16)
            st = self._stage[:e - s]
            if seg.is_cuda:
                from . import ops
                ops.cast_f32_bf16(seg, st)
            else:
                st.copy_(seg)
            h = dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            h.wait()                                               # stream-side on CUDA; the staging buffer is reused by the next bucket
            if seg.is_cuda:
                from . import ops
                ops.cast_bf16_f32(st, seg)
            else:
                seg.copy_(st)
            return None
        return dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def ready(self, upto: int, events=()) -> None:
        """Gradients below ``upto`` are final once the current stream and ``events`` (recorded on the producer's other
        streams) have passed: the collective stream waits for them, the producer does not stall."""
        while self.next < len(self.buckets) and self.buckets[self.next][1] <= upto:
            s, e = self.buckets[self.next]
            self.next += 1
            if self.side is not None:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                with torch.cuda.stream(self.side):
                    self.side.wait_event(ev)
                    for x in events:
                        self.side.wait_event(x)
                    if self.timing:
                        t0 = torch.cuda.Event(enable_timing=True)
                        t0.record(self.side)
                    h = self._reduce(s, e)
                    if h is not None:
                        self.handles.append(h)
                    if self.on_bucket is not None or self.timing:
                        if h is not None:
                            h.wait()                               # stream-side wait (the side stream, not the host)
                    if self.timing:
                        t1 = torch.cuda.Event(enable_timing=True)
                        t1.record(self.side)
                        self._ev.append((t0, t1))
                    if self.on_bucket is not None:
                        self.on_bucket(e)
            else:
                h = self._reduce(s, e)
                if h is not None:
                    self.handles.append(h)
                if self.on_bucket is not None:
                    if h is not None:
                        h.wait()
                    self.on_bucket(e)

    def finish(self) -> None:
        if self.timing and self.side is not None:
            self._bwd_end = torch.cuda.Event(enable_timing=True)
            self._bwd_end.record(torch.cuda.current_stream())     # the backward pass (on the producer's stream) ends here
        self.ready(self.flat.numel())
        for h in self.handles:
            h.wait()
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
        self.handles = []
        if self.timing and self.side is not None and self._ev:
            torch.cuda.current_stream().synchronize()
            comm = sum(a.elapsed_time(b) for a, b in self._ev)
            exposed = max(0.0, self._bwd_end.elapsed_time(self._ev[-1][1]))
            self._stats["steps"] += 1
            self._stats["comm_ms"] += comm
            self._stats["exposed_ms"] += min(exposed, comm)

    def stats(self, reset: bool = False) -> dict:
        """Per-step averages since the last reset: time inside the collectives / time the step waited for them after the backward pass."""
        n = max(1, self._stats["steps"])
        out = dict(steps=self._stats["steps"], comm_ms=self._stats["comm_ms"] / n, comm_exposed_ms=self._stats["exposed_ms"] / n,
                   buckets=len(self.buckets), bucket_mb=round(4e-6 * max(b - a for a, b in self.buckets), 1), bucket_dtype=self.bucket_dtype)
        if reset:
            self._stats = dict(steps=0, comm_ms=0.0, exposed_ms=0.0)
        return out


def rank_seed(base_seed: int, rank: int) -> int:
    """Dropout base seed of a rank: a splitmix64 hash of (base_seed, rank), 26 bits so that the model's per-step seed
    (base_seed * 1000003 + step * 7919, + small per-layer offsets) stays injective in (base_seed, step, offset).  Rank 0 keeps
    ``base_seed`` (the single-process stream)."""
    if rank == 0:
        return base_seed
    z = ((base_seed & 0xFFFFFFFF) | (rank << 32)) + 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    z ^= z >> 31
    return int(z & 0x3FFFFFF) | 1


class GradSync:
    """Glue between a finalized model's engine and ``BucketedAllReduce``."""

    def __init__(self, model, bucket_elems: int = 32 << 20, group=None):
        self.model = model
        import os
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        st = model.store
        # MART_FORCE_PG=1: run the bucketed collectives even in a 1-rank group (exercises the RCCL path on a 1-GPU box)
        on = self.world > 1 or (dist.is_initialized() and os.environ.get("MART_FORCE_PG") == "1")
        self.reducer = BucketedAllReduce(st.grad, st.buckets(bucket_elems), group) if on else None
        if self.reducer is not None:
            model.engine.grad_ready_async = self.reducer.ready       # (offset, events): no stream joins in the backward pass
        self.group = group
        if dist.is_initialized() and (self.world > 1 or on):
            # Replicas start from rank 0's weights (what PL-DDP's module broadcast does): resize_token_embeddings draws the new
            # [ENTITY_i] / [RELATION_j] rows from the per-process RNG, so without this the ranks would train different networks
            # that never re-synchronise (only gradients are all-reduced).
            dist.broadcast(st.master, src=0, group=group)
            st.refresh_shadows()
            # ... and draw DIFFERENT dropout masks: the counter-based dropout seed is derived from base_seed, identical on
            # every rank by construction; mix the rank in (rank 0 keeps the single-process stream)
            rank = dist.get_rank(group)
            if not hasattr(model, "base_seed"):
                raise RuntimeError("GradSync: the model has no base_seed attribute; every rank would draw the same dropout masks")
            if rank and not getattr(model, "_rank_seeded", False):
                model.base_seed = rank_seed(int(model.base_seed), rank)
                model._rank_seeded = True

    def broadcast_optimizer(self, optimizer) -> None:
        """Resume path: AdamW moments and the step count follow rank 0 as well."""
        if not dist.is_initialized() or self.world == 1:
            return
        for name in ("m", "v"):
            t = getattr(optimizer, name, None)
            if torch.is_tensor(t):
                dist.broadcast(t, src=0, group=self.group)
        steps = torch.tensor([int(getattr(optimizer, "steps", 0))], device=self.model.store.master.device, dtype=torch.int64)
        dist.broadcast(steps, src=0, group=self.group)
        if hasattr(optimizer, "steps"):
            optimizer.steps = int(steps)

    def begin(self) -> None:
        if self.reducer is not None:
            self.reducer.begin()

    def finish(self) -> None:
        if self.reducer is not None:
            self.reducer.finish()

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, local_rank, world) from torchrun's environment; initialises the default process group when world > 1."""
    import os
    rank, local, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    # test hooks (tests/test_two_ranks_one_gpu.py): RCCL refuses two ranks on one device, so the N > 1 control flow is
    # exercised on a 1-GPU box with every rank on MART_DEVICE_INDEX and the gloo backend moving the CUDA tensors
    if os.environ.get("MART_DEVICE_INDEX") is not None:
        local = int(os.environ["MART_DEVICE_INDEX"])
    backend = backend or os.environ.get("MART_DIST_BACKEND")
    if (world > 1 or os.environ.get("MART_FORCE_PG") == "1") and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, local, world


def all_gather_ranks(ranks, group=None):
    """Concatenate per-rank numpy rank arrays (eval: SURVEY 8(e))."""
    import numpy as np
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return ranks
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, ranks, group=group)
    return np.concatenate(out)
