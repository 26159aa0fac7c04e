"""Fused AdamW over the flat parameter buffer + the linear warmup/decay schedule
(reference: lit_models/transformer.py:224-241; torch.optim.AdamW and HF get_linear_schedule_with_warmup semantics)."""
from __future__ import annotations

import os

import torch

from . import ops
from .params import NO_DECAY


class FusedAdamW(torch.optim.Optimizer):
    """One multi-tensor HIP launch per step: fp32 master/m/v update + bf16 shadow refresh, then W^T shadows.

    ``param_groups`` mirrors the reference's two groups (decay 0.01 / 0 by the substring rule) for inspection and
    for the scheduler's ``lr`` plumbing; tensors that never receive a gradient in the reference (dead pooler /
    post-layernorm: grad None => torch skips them entirely) are left untouched.
    """

    def __init__(self, model, lr=5e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, grad_scale: float = 1.0):
        self.model = model
        store = model.store
        named = dict(model.named_parameters())
        decay = [p for n, p in named.items() if p.requires_grad and not any(nd in n for nd in NO_DECAY)]
        no_decay = [p for n, p in named.items() if p.requires_grad and any(nd in n for nd in NO_DECAY)]
        groups = [{"params": decay, "weight_decay": weight_decay}, {"params": no_decay, "weight_decay": 0}]
        super().__init__(groups, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.m = torch.zeros_like(store.master)
        self.v = torch.zeros_like(store.master)
        self.steps = 0
        self.grad_scale = grad_scale
        self._store_id = id(store)
        self._stream = None
        self._streaming = False
        # MART_ASYNC_STEP=1: step-boundary work that the NEXT forward pass does not need -- the gradient zero-fill and the refresh of the transposed
        # weight shadows (read by data-gradient GEMMs only) -- stays on the optimizer stream; the backward pass joins it (FlatStore.join_pending).
        # Off by default: measured within noise of the in-order form (87.24 / 87.50 vs 87.47 / 87.36 ms: the un-profiled step boundary has no idle
        # time to recover, docs/LAB_r01-r05.md section 4.3), and with it ``zero_grad()`` returns before ``p.grad`` reads as zero on the caller's stream.
        self.async_step = os.environ.get("MART_ASYNC_STEP", "0") == "1"
        # fused_zero_grad: the update kernel writes 0 back to the gradient elements it has just consumed, and the next ``zero_grad()`` finds the buffer
        # clean (FlatStore.grad_clean) and skips its 0.94 GB fill -- 0.13-0.19 ms at the head of every step on the main queue, moved under the backward
        # pass where the update ranges already run.  torch semantics differ in ONE observable way: ``p.grad`` reads zero after ``optimizer.step()``
        # instead of keeping the step's gradients until the next ``zero_grad()``.  The Trainer (which owns both calls, lit_models hooks see the
        # gradients between backward and step as in PL) turns it on; a bare ``FusedAdamW`` keeps torch's behaviour (MART_FUSED_ZERO_GRAD=0: off everywhere).
        self.fused_zero_grad = False

    def _side(self):
        if self._stream is None:
            self._stream = torch.cuda.Stream(priority=int(os.environ.get("MART_OPT_PRIO", "0")))
        return self._stream

    def zero_grad(self, set_to_none: bool = False):
        """With ``async_step`` the 0.94 GB fill runs on the optimizer stream, next to the forward pass (which never touches the gradient buffer); whatever
        writes gradients waits for it first (FlatStore.join_pending: the model's forward call joins before returning, and every
        backward kernel is enqueued later).  Default (``MART_ASYNC_STEP=0``): the plain in-order fill."""
        store = self.model.store
        if getattr(store, "grad_clean", False):
            return                                                       # zeroed by the last update (fused_zero_grad) and not written since
        if not (self.async_step and store.grad.is_cuda):
            store.zero_grad()
            return
        st = self._side()
        st.wait_stream(torch.cuda.current_stream())                      # readers / writers of the gradients enqueued so far
        with torch.cuda.stream(st):
            store.grad.zero_()
        store.pending(st)

    # ---- streamed step: the update of a parameter range is launched as soon as its gradients are final (the backward
    # pass reports "everything below flat offset X is final", layout order == completion order), on its own stream, under
    # the rest of the backward pass: AdamW is HBM-bound, the backward GEMMs are not, and the ~1.4 ms of the update
    # otherwise run alone at the end of every step.  The weights of a finished range are not read again in this step.
    def begin_step(self) -> None:
        """Call before the backward pass whose gradients this step consumes; ``ready`` / ``step`` then finish it."""
        store = self.model.store
        if id(store) != self._store_id:
            raise RuntimeError("the model's parameter storage was rebuilt (resize/.to()) after the optimizer was created")
        self.steps += 1
        g0 = self.param_groups[0]
        b1, b2 = g0["betas"]
        self._hp = dict(lr=float(g0["lr"]), beta1=b1, beta2=b2, eps=g0["eps"], weight_decay=float(g0["weight_decay"]),
                        bc1=1.0 - b1 ** self.steps, bc2=1.0 - b2 ** self.steps, grad_scale=self.grad_scale, zero_grad=bool(self.fused_zero_grad))
        if getattr(self, "_ends_key", None) is not store.chunks:          # host copy of the chunk ends (rebuilt with the table)
            c = store.chunks.cpu()
            self._ends = (c[:, 0].long() + c[:, 1].long()).tolist()
            self._ends_key = store.chunks
        self._cursor = 0
        self._streaming = True

    def _launch(self, upto_chunk: int) -> None:
        store = self.model.store
        if upto_chunk > self._cursor:
            ops.adamw(master=store.master, grad=store.grad, m=self.m, v=self.v, shadow=store.shadow, shadow_f16=store.shadow_h,
                      chunks=store.chunks[self._cursor:upto_chunk],
                      n_chunks=upto_chunk - self._cursor, **self._hp)
            self._cursor = upto_chunk

    @torch.no_grad()
    def ready(self, upto: int, events=()) -> None:
        """Gradients below flat offset ``upto`` are final once ``events`` (and the current stream) have passed."""
        if not getattr(self, "_streaming", False):
            return
        import bisect
        j = bisect.bisect_right(self._ends, upto)
        if j <= self._cursor:
            return
        if not self.m.is_cuda:
            self._launch(j)
            return
        self._side()
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._stream.wait_event(ev)
        for e in events:
            self._stream.wait_event(e)
        with torch.cuda.stream(self._stream):
            self._launch(j)

    @torch.no_grad()
    def step(self, closure=None):
        store = self.model.store
        if not getattr(self, "_streaming", False):
            self.begin_step()
        self.ready(store.total)                                           # whatever the backward pass did not release
        self._launch(store.n_chunks)
        self._streaming = False
        if self.fused_zero_grad:
            store.grad_clean = True                                       # every chunk's gradients were zeroed by its update; nothing else is ever written
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)       # master + bf16 / fp16 shadows are final for the next forward pass
        if self.async_step and self._stream is not None and store.grad.is_cuda:
            with torch.cuda.stream(self._stream):
                store.refresh_transposed()                               # W^T shadows: only the NEXT backward pass reads them
            store.pending(self._stream)
        else:
            store.refresh_transposed()

    def state_dict(self):
        return {"m": self.m, "v": self.v, "steps": self.steps, "lr": self.param_groups[0]["lr"]}

    def load_state_dict(self, sd):
        self.m.copy_(sd["m"]); self.v.copy_(sd["v"]); self.steps = int(sd["steps"])
        for g in self.param_groups:
            g["lr"] = sd["lr"]


class TorchOptimizerOnStore:
    """Any other ``torch.optim`` class by name, as the reference allows (lit_models/base.py:31 ``getattr(torch.optim, optimizer)``;
    lit_models/transformer.py:224-232: the same two weight-decay groups, ``lr``, ``eps=1e-8`` -- so, as there, only classes that take ``eps``).
    The update is torch's own (its kernels, on the fp32 master views: ``p`` / ``p.grad`` are views of FlatStore.master / .grad); afterwards the
    bf16 / fp16 / W^T operand shadows follow.  Tensors that never receive a gradient in the reference (grad None: torch skips them -- no update,
    no weight decay) are left out, like FusedAdamW's chunk table does.  Not the fast path: AdamW is the fused kernel."""

    def __init__(self, model, name: str, lr=5e-5, eps=1e-8, weight_decay=0.01, extra_dead=()):
        cls = getattr(torch.optim, name)                       # AttributeError for an unknown name, as the reference
        store = model.store
        dead = tuple(store.dead) + tuple(extra_dead)
        named = {n: p for n, p in model.named_parameters() if p.requires_grad and not (dead and n.startswith(dead))}
        decay = [p for n, p in named.items() if not any(nd in n for nd in NO_DECAY)]
        no_decay = [p for n, p in named.items() if any(nd in n for nd in NO_DECAY)]
        self.inner = cls([{"params": decay, "weight_decay": weight_decay}, {"params": no_decay, "weight_decay": 0}], lr=lr, eps=eps)
        self.model, self.grad_scale, self._store_id = model, 1.0, id(store)

    @property
    def param_groups(self):
        return self.inner.param_groups

    def zero_grad(self, set_to_none: bool = False):
        self.model.store.zero_grad()                           # the gradient buffer is persistent (kernels accumulate into it): never None

    @torch.no_grad()
    def step(self, closure=None):
        store = self.model.store
        if id(store) != self._store_id:
            raise RuntimeError("the model's parameter storage was rebuilt (resize/.to()) after the optimizer was created")
        if self.grad_scale != 1.0:
            store.grad.mul_(self.grad_scale)                   # 1 / world (DDP mean) and 1 / accumulation windows
        self.inner.step()
        store.refresh_shadows()

    def state_dict(self):
        return self.inner.state_dict()

    def load_state_dict(self, sd):
        self.inner.load_state_dict(sd)


class LinearWarmupSchedule:
    """lambda(t) = t/max(1,w) for t<w else max(0,(T-t)/max(1,T-w)); w may be fractional (0.1*T)."""

    def __init__(self, optimizer, num_warmup_steps, num_training_steps, last_epoch: int = -1):
        self.optimizer = optimizer
        self.w, self.T = num_warmup_steps, num_training_steps
        self.base_lrs = [g["lr"] for g in optimizer.param_groups]
        self.last_epoch = last_epoch
        self.step()                                                    # like torch's LambdaLR: lr(0) applied at construction

    def lr_lambda(self, t: int) -> float:
        if t < self.w:
            return float(t) / float(max(1, self.w))
        return max(0.0, float(self.T - t) / float(max(1, self.T - self.w)))

    def step(self):
        self.last_epoch += 1
        f = self.lr_lambda(self.last_epoch)
        for g, b in zip(self.optimizer.param_groups, self.base_lrs):
            g["lr"] = b * f

    def get_last_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]

    def state_dict(self):
        return {"last_epoch": self.last_epoch}

    def load_state_dict(self, sd):
        self.last_epoch = sd["last_epoch"] - 1
        self.step()
