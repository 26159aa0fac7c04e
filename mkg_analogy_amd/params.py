"""Flat parameter storage for the MKGformer path.

All 451 trainable tensors live in ONE fp32 master buffer (the nn.Parameters are views into it), with
  * one fp32 gradient buffer of the same layout (``p.grad`` are views; backward kernels accumulate in place,
    RCCL all-reduces contiguous slices of it),
  * a bf16 shadow of the same layout that the MFMA GEMMs read (refreshed by the fused AdamW kernel), and
  * a bf16 buffer of transposed GEMM weights W^T (so data-gradient GEMMs are NT as well).
Layout order = order in which backward finishes the gradients (head, layer 11 .. 0, embeddings, tied word
embedding last) so that gradient buckets can be reduced while backward is still running.  Every tensor starts on
a 256-element boundary; q/k/v projections of a layer are adjacent so the shadow holds a ready [3H,H] fused matrix.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

ALIGN = 256
NO_DECAY = ("bias", "LayerNorm.weight")          # lit_models/transformer.py:225 (substring match, reproduced as is)


@dataclass
class Slot:
    name: str
    shape: Tuple[int, ...]
    offset: int
    numel: int


def layout_order(n_layers: int) -> List[str]:
    o: List[str] = []
    h = "cls.predictions.transform."
    o += [h + "dense.weight", h + "dense.bias", h + "LayerNorm.weight", h + "LayerNorm.bias"]
    for l in reversed(range(n_layers)):
        t = f"unimo.encoder.text_layer.{l}."
        o += [t + f"attention.self.{n}.weight" for n in ("query", "key", "value")]
        o += [t + f"attention.self.{n}.bias" for n in ("query", "key", "value")]
        o += [t + "attention.self.adaptive_weight.0", t + "attention.self.adaptive_weight.1"]
        o += [t + "attention.output.dense.weight", t + "attention.output.dense.bias",
              t + "attention.output.LayerNorm.weight", t + "attention.output.LayerNorm.bias",
              t + "intermediate.dense.weight", t + "intermediate.dense.bias",
              t + "intermediate.fusion_dense.weight", t + "intermediate.fusion_dense.bias",
              t + "output.dense.weight", t + "output.dense.bias", t + "output.LayerNorm.weight", t + "output.LayerNorm.bias"]
        if l == 0:
            # the text stream ends here: its embedding tables (the tied 32 M-element word embedding among them) are final once text layer 0 and the
            # embedding backward are through -- typically while the vision stream is still in its last layers -- so their all-reduce bucket / AdamW range
            # is released then (engine.backward) instead of after the whole pass (round 6; they used to sit at the very end of the buffer)
            u = "unimo."
            o += [u + "text_embeddings.LayerNorm.weight", u + "text_embeddings.LayerNorm.bias",
                  u + "text_embeddings.position_embeddings.weight", u + "text_embeddings.token_type_embeddings.weight",
                  "cls.predictions.bias", u + "text_embeddings.word_embeddings.weight"]
        v = f"unimo.encoder.vision_layers.{l}."
        o += [v + f"self_attn.{n}.weight" for n in ("q_proj", "k_proj", "v_proj")]
        o += [v + f"self_attn.{n}.bias" for n in ("q_proj", "k_proj", "v_proj")]
        o += [v + "self_attn.out_proj.weight", v + "self_attn.out_proj.bias", v + "layer_norm1.weight", v + "layer_norm1.bias",
              v + "mlp.fc1.weight", v + "mlp.fc1.bias", v + "mlp.fc2.weight", v + "mlp.fc2.bias",
              v + "layer_norm2.weight", v + "layer_norm2.bias"]
    u = "unimo."
    o += [u + "vision_pre_layrnorm.weight", u + "vision_pre_layrnorm.bias",
          u + "vision_embeddings.class_embedding", u + "vision_embeddings.position_embedding.weight",
          u + "vision_embeddings.patch_embedding.weight",
          # never receive a gradient in the reference (dead pooler :748, unused post layernorm :683)
          u + "vision_post_layernorm.weight", u + "vision_post_layernorm.bias",
          u + "text_pooler.dense.weight", u + "text_pooler.dense.bias"]
    return o


DEAD = ("unimo.vision_post_layernorm.", "unimo.text_pooler.")


def text_f16_weight(name: str) -> bool:
    """GEMM weights of the MKGformer text layers: the tensors that get an fp16 forward shadow (engine.text_f16)."""
    return name.startswith("unimo.encoder.text_layer.") and name.endswith(".weight") and "LayerNorm" not in name


PACK_WITH_PREV = ("adaptive_weight.1",)           # shares the 256-element slot of adaptive_weight.0 (contiguous [2])


def gemm_weight_names(n_layers: int) -> List[Tuple[str, Tuple[str, ...]]]:
    """(key, member weight names) of every matrix that needs a transposed shadow; members are adjacent in the flat
    layout and are treated as one [sum(out), in] matrix."""
    g: List[Tuple[str, Tuple[str, ...]]] = [("head", ("cls.predictions.transform.dense.weight",))]
    for l in range(n_layers):
        t = f"unimo.encoder.text_layer.{l}."
        g += [(f"t{l}.qkv", tuple(t + f"attention.self.{n}.weight" for n in ("query", "key", "value"))),
              (f"t{l}.ao", (t + "attention.output.dense.weight",)),
              (f"t{l}.int", (t + "intermediate.dense.weight",)),
              (f"t{l}.fus", (t + "intermediate.fusion_dense.weight",)),
              (f"t{l}.out", (t + "output.dense.weight",))]
        v = f"unimo.encoder.vision_layers.{l}."
        g += [(f"v{l}.qkv", tuple(v + f"self_attn.{n}.weight" for n in ("q_proj", "k_proj", "v_proj"))),
              (f"v{l}.o", (v + "self_attn.out_proj.weight",)),
              (f"v{l}.fc1", (v + "mlp.fc1.weight",)),
              (f"v{l}.fc2", (v + "mlp.fc2.weight",))]
    return g


class FlatStore:
    def __init__(self, named_params: Dict[str, torch.nn.Parameter], n_layers: int, device: torch.device,
                 order: Optional[List[str]] = None, gemm_groups=None, dead: Sequence[str] = DEAD, f16_weight=text_f16_weight):
        """``order`` / ``gemm_groups`` / ``dead`` / ``f16_weight`` default to the MKGformer layout; other backbones (FLAVA) pass their own."""
        order = layout_order(n_layers) if order is None else order
        gemm_groups = gemm_weight_names(n_layers) if gemm_groups is None else gemm_groups
        self.dead = tuple(dead)
        missing = set(named_params) - set(order)
        extra = set(order) - set(named_params)
        assert not missing and not extra, f"parameter set mismatch: missing {sorted(missing)[:4]} extra {sorted(extra)[:4]}"
        self.device = device
        self.slots: Dict[str, Slot] = {}
        off = 0
        for name in order:
            p = named_params[name]
            n = p.numel()
            if name.endswith(PACK_WITH_PREV):
                prev = self.slots[name[:-1] + "0"]
                self.slots[name] = Slot(name, tuple(p.shape), prev.offset + prev.numel, n)
                continue
            self.slots[name] = Slot(name, tuple(p.shape), off, n)
            off += ((n + ALIGN - 1) // ALIGN) * ALIGN
        self.total = off
        self.master = torch.zeros(off, device=device, dtype=torch.float32)
        self.grad = torch.zeros(off, device=device, dtype=torch.float32)
        self.shadow = torch.zeros(off, device=device, dtype=torch.bfloat16)
        # fp16 shadow of the same layout, maintained for the tensors ``f16_weight`` selects (forward weights of the text stream: fp16 rounds 8x
        # finer than bf16 at the same MFMA rate, and N(0, 0.02)-scale weights sit inside its range); the other positions are never read
        self.f16_weight = f16_weight
        self.shadow_h = torch.zeros(off, device=device, dtype=torch.float16)
        with torch.no_grad():
            for name, s in self.slots.items():
                p = named_params[name]
                view = self.master[s.offset:s.offset + s.numel].view(s.shape)
                view.copy_(p.data.to(device=device, dtype=torch.float32))
                p.data = view
                p.grad = self.grad[s.offset:s.offset + s.numel].view(s.shape)
        # transposed GEMM weights
        self.tslots: Dict[str, Tuple[int, int, int]] = {}        # key -> (offset, rows(out), cols(in))
        toff = 0
        table = []
        for key, names in gemm_groups:
            first = self.slots[names[0]]
            rows = sum(self.slots[n].shape[0] for n in names)
            cols = first.shape[1]
            exp = first.offset
            for n in names:                                       # adjacency check
                assert self.slots[n].offset == exp, f"{n} not adjacent"
                exp += self.slots[n].numel
                assert self.slots[n].numel % ALIGN == 0
            self.tslots[key] = (toff, rows, cols)
            table.append([first.offset, toff, rows, cols])
            toff += rows * cols
        self.shadow_t = torch.zeros(toff, device=device, dtype=torch.bfloat16)
        self.ttable = torch.tensor(table, dtype=torch.int64, device=device)
        # AdamW chunk table: (start, len, decay_flag); dead tensors are excluded (grad None in the reference => untouched)
        self.order = order
        self.version = 0                                           # bumped whenever the master weights change (derived operand caches)
        self.grad_clean = True                                     # the gradient buffer is all-zero (FusedAdamW.fused_zero_grad / zero_grad)
        self.rebuild_chunks()
        self.refresh_shadows()

    def rebuild_chunks(self, extra_dead: Sequence[str] = ()) -> None:
        """AdamW chunk table.  Tensors that never receive a gradient in the reference (grad None => torch.optim skips them:
        no update, no weight decay) are left out; ``extra_dead`` adds run-mode specific ones (e.g. adaptive weights when
        pre-training without sep_idx)."""
        device = self.device
        chunks = []
        CH = 1 << 16
        dead = tuple(self.dead) + tuple(extra_dead)
        for name in self.order:
            if name.endswith(PACK_WITH_PREV):
                continue
            if dead and name.startswith(dead):
                continue
            s = self.slots[name]
            n = s.numel + (1 if name.endswith("adaptive_weight.0") else 0)
            decay = 0 if any(nd in name for nd in NO_DECAY) else 1
            if self.f16_weight is not None and self.f16_weight(name):
                decay |= 2                                     # flag bit 1: AdamW refreshes the fp16 shadow of this chunk too
            for c0 in range(0, n, CH):
                chunks.append([s.offset + c0, min(CH, n - c0), decay])
        self.chunks = torch.tensor(chunks, dtype=torch.int32, device=device)
        self.n_chunks = len(chunks)

    # ------------------------------------------------------------------ views
    def m(self, name: str) -> torch.Tensor:
        s = self.slots[name]
        return self.master[s.offset:s.offset + s.numel].view(s.shape)

    def g(self, name: str) -> torch.Tensor:
        if getattr(self, "_pending", None):
            self.join_pending()                      # an off-stream zero-fill may still be in flight (optim.FusedAdamW.zero_grad)
        s = self.slots[name]
        self.grad_clean = False                      # handed out to a writer (every kernel's gradient destination comes from here or from fused())
        return self.grad[s.offset:s.offset + s.numel].view(s.shape)

    def w(self, name: str) -> torch.Tensor:
        """bf16 shadow of a (2-D) tensor."""
        s = self.slots[name]
        return self.shadow[s.offset:s.offset + s.numel].view(s.shape)

    def h(self, *names: str) -> torch.Tensor:
        """fp16 shadow of one (or several adjacent) 2-D weights selected by ``f16_weight``."""
        assert all(self.f16_weight(n) for n in names), names
        return self.fused(list(names), self.shadow_h)

    def fused(self, names: Sequence[str], buf: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Adjacent tensors as one [sum(rows), ...] view of ``buf`` (default: bf16 shadow)."""
        buf = self.shadow if buf is None else buf
        if buf is self.grad:
            self.grad_clean = False
        first = self.slots[names[0]]
        rows = sum(self.slots[n].shape[0] for n in names)
        tail = first.shape[1:]
        n = rows
        for t in tail:
            n *= t
        # members are 256-aligned multiples, so the concatenation is gap-free (checked for GEMM groups in __init__)
        return buf[first.offset:first.offset + n].view((rows,) + tuple(tail))

    def wt(self, key: str) -> torch.Tensor:
        """bf16 transposed shadow [in, sum(out)] of a GEMM weight group."""
        if getattr(self, "_pending", None):
            self.join_pending()                      # an off-stream refresh may still be in flight (optim.FusedAdamW.step)
        off, rows, cols = self.tslots[key]
        return self.shadow_t[off:off + rows * cols].view(cols, rows)

    def owns(self, named_params: Dict[str, torch.nn.Parameter]) -> bool:
        base = self.master.data_ptr()
        for name, p in named_params.items():
            s = self.slots.get(name)
            if s is None or tuple(p.shape) != s.shape or p.data_ptr() != base + 4 * s.offset:
                return False
        return len(named_params) == len(self.slots)

    def bind(self, model: torch.nn.Module) -> None:
        """Remember where every stored parameter hangs in ``model`` (module._parameters dict, key, object, address), so that the per-call
        ownership check (:meth:`still_bound`) is 450 dict lookups instead of a ``named_parameters()`` walk of the module tree (0.4-1 ms, eight
        times per training step)."""
        by_id = {id(p): n for n, p in model.named_parameters(remove_duplicate=False) if n in self.slots}
        base = self.master.data_ptr()
        bound = []
        for mod in model.modules():
            for key, p in mod._parameters.items():
                n = by_id.get(id(p)) if p is not None else None
                if n is not None:
                    bound.append((mod._parameters, key, p, base + 4 * self.slots[n].offset))
        self._bound = bound

    def still_bound(self) -> bool:
        """True while every parameter object recorded by :meth:`bind` is still the one its module holds and still lives at its slot of the flat
        buffer (a ``.to()`` / ``.float()`` / ``.data =`` or a re-assigned ``nn.Parameter`` fails this; the caller then falls back to the
        full walk and, if needed, rebuilds the store).  Parameters ADDED to the model later are not part of the path this engine computes."""
        b = getattr(self, "_bound", None)
        if not b:
            return False
        for d, key, p, ptr in b:
            if d.get(key) is not p or p.data_ptr() != ptr:
                return False
        return True

    def refresh_shadows(self) -> None:
        """master -> bf16 shadow and W^T shadow (after loading weights; AdamW keeps the first one fresh itself)."""
        from . import ops
        self.join_pending()                          # an off-stream W^T refresh of the previous optimizer step may still be in flight
        self.version += 1
        ops.cast_f32_bf16(self.master, self.shadow)
        ops.cast_f32_f16(self.master, self.shadow_h)
        ops.transpose_table(self.shadow, self.shadow_t, self.ttable, self.ttable.shape[0])

    def refresh_transposed(self) -> None:
        """After the fused AdamW step (which rewrote master + bf16 shadow)."""
        from . import ops
        self.version += 1
        ops.transpose_table(self.shadow, self.shadow_t, self.ttable, self.ttable.shape[0])

    def zero_grad(self) -> None:
        self.join_pending()
        self.grad.zero_()
        self.grad_clean = True

    def touch_grad(self) -> None:
        """For writers that go to ``store.grad`` / ``p.grad`` directly (g() and fused(.., grad) mark the buffer themselves): no longer all-zero."""
        self.grad_clean = False

    # ---- work enqueued on another stream that the gradient writers / the readers of the transposed shadows must not overtake
    def pending(self, stream) -> None:
        """``stream`` carries a gradient zero-fill or a W^T refresh issued off the compute stream (optim.FusedAdamW)."""
        ev = torch.cuda.Event()
        ev.record(stream)
        self._pending = getattr(self, "_pending", []) + [ev]

    def join_pending(self) -> None:
        """The current stream waits for everything ``pending`` announced (called by the model's forward, after its last kernel: every
        backward kernel is enqueued after it)."""
        for ev in getattr(self, "_pending", []):
            torch.cuda.current_stream().wait_event(ev)
        self._pending = []

    def buckets(self, bucket_elems: int = 16 << 20) -> List[Tuple[int, int]]:
        """Contiguous (start, end) slices of the gradient buffer in backward-completion order."""
        out = []
        s = 0
        while s < self.total:
            e = min(self.total, s + bucket_elems)
            out.append((s, e))
            s = e
        return out
