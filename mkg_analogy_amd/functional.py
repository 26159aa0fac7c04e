"""Autograd nodes that hang the HIP engine into ``loss.backward()``.

Coarse on purpose: one node for the whole encoder + head transform, one for the scoring slice of the tied decoder,
one each for the two loss terms.  Parameter gradients are NOT returned through autograd; the kernels accumulate
into ``FlatStore.grad`` (which ``p.grad`` views), the way fused wgrad accumulation is usually done.
"""
from __future__ import annotations

import weakref
from typing import Optional

import torch

from . import ops

BF, F32 = torch.bfloat16, torch.float32
WORD, BIAS = "unimo.text_embeddings.word_embeddings.weight", "cls.predictions.bias"


class _MKGformerFn(torch.autograd.Function):
    """trans_hidden_states = head_transform(encoder(...)).  UnimoForMaskedLM.forward, modeling_unimo.py:848-893."""

    @staticmethod
    def forward(ctx, anchor, engine, input_ids, attention_mask, token_type_ids, pixel_values, sep_idx, train, seed, holder,
                image_table=None, image_index=None, rows=None):
        kw = {} if rows is None else dict(rows=rows)          # last-layer row subset (engine.UnimoEngine.forward)
        trans, transb, sv = engine.forward(input_ids, attention_mask, token_type_ids, pixel_values, sep_idx, train, seed,
                                           image_table=image_table, image_index=image_index, **kw)
        ctx.engine, ctx.sv = engine, sv
        holder["trans_bf16"] = transb
        return trans                                 # [B, L, H]; with ``rows``: the compact [B * nr, H] tensor (RowSubset / _DenseRowsFn below)

    @staticmethod
    def backward(ctx, dtrans):
        sv, ctx.sv = ctx.sv, None
        if sv is None:
            raise RuntimeError("MKGformer backward called twice (activations are freed after the first pass)")
        ctx.engine.backward(sv, dtrans)
        return (None,) * 13


class RowSubset:
    """The compact tensors of a row-subset pass (engine.forward(rows=...)): ``f32`` [B * nr, H] (the autograd output of the engine node), ``bf16`` its
    bf16 copy, ``rows`` int32 [B, nr] flat ids b * L + position, ``mask_index`` (optional) int32 [B] compact indices of the [MASK] rows (slot 0).
    Carried on the dense [B, L, H] tensor the model returns (attribute ``_mart_rows``) so that the scoring head and the relaxation loss read -- and
    send their gradients to -- the compact rows: no [B, L, H] zero-fill, scatter, add or gather between the loss and the engine's backward pass."""

    def __init__(self, f32, bf16, rows, L, mask_index=None, mask_row=None, mask_token=None):
        self.f32, self.bf16, self.rows, self.L, self.mask_index = f32, bf16, rows, int(L), mask_index
        self.mask_row, self.mask_token = mask_row, mask_token

    def lookup(self, flat_rows: torch.Tensor) -> torch.Tensor:
        """compact indices of flat row ids (first slot that names the row; a row outside the promise: flagged for check_status)."""
        out = torch.empty(flat_rows.numel(), device=flat_rows.device, dtype=torch.int32)
        ops.rows_lookup(flat_rows.contiguous(), self.rows, self.L, out, status=_status(flat_rows.device))
        return out


class _DenseRowsFn(torch.autograd.Function):
    """[B, L, H] view of a compact row-subset tensor: the promised rows, NaN elsewhere (a consumer that breaks the promise -- reduces over all of
    trans_hidden_states, reads another position -- gets NaN instead of silently wrong numbers; the reference computes every row).  Backward: the
    gradient rows of the promised positions (a position named twice: taken once, by its first slot)."""

    @staticmethod
    def forward(ctx, compact, rows, B, L):
        H = compact.shape[-1]
        dense = torch.empty((B, L, H), device=compact.device, dtype=F32)
        ops.rows_dense(compact, rows, B, L, dense.view(-1, H))
        ctx.rows, ctx.shape = rows, compact.shape
        return dense

    @staticmethod
    def backward(ctx, d):
        rows = ctx.rows
        out = torch.empty(ctx.shape, device=d.device, dtype=F32)
        ops.gather_rows_first_f32(d.contiguous().view(-1, d.shape[-1]), rows.view(-1), int(rows.shape[1]), out)
        return out, None, None, None


_UNIQUE_OK: dict = {}      # id(tensor) -> (weak reference to that tensor object, the _version it was checked at)


def _require_unique(ids: torch.Tensor) -> None:
    """The deterministic weight-gradient reduction adds each scored vocabulary row with a plain read-modify-write
    (mart_gemm_tn with a workspace: out_rows must not repeat).  Ranges built by LazyRows from a slice are unique by construction
    (``_mart_unique``); any other id tensor is checked (one device sync) once per tensor OBJECT and in-place version -- never per
    address: the caching allocator hands the address of a freed id tensor to the next one."""
    if getattr(ids, "_mart_unique", False):
        return
    hit = _UNIQUE_OK.get(id(ids))
    if hit is not None and hit[0]() is ids and hit[1] == ids._version:
        return
    if int(torch.unique(ids).numel()) != ids.numel():
        raise ValueError("scored vocabulary ids must be unique (the tied-embedding gradient rows are scattered without atomics)")
    if len(_UNIQUE_OK) > 64:
        for k in [k for k, (r, _) in _UNIQUE_OK.items() if r() is None]:
            del _UNIQUE_OK[k]
    _UNIQUE_OK[id(ids)] = (weakref.ref(ids), ids._version)


class _ScoreFn(torch.autograd.Function):
    """logits[rows][:, ids] of the tied decoder: trans[rows] @ word_emb[ids]^T + bias[ids]
    (modeling_unimo.py:958 restricted to what lit_models/transformer.py:75-95,131-160 actually read)."""

    @staticmethod
    def forward(ctx, trans, transb, rows, ids, store, word_name=WORD, bias_name=BIAS, head_split=True):
        R, A = rows.numel(), ids.numel()
        if trans.requires_grad:
            _require_unique(ids)
        out = torch.empty((R, A), device=trans.device, dtype=F32)
        if head_split:
            # two-term bf16 splits of the f32 rows on both sides (K' = 3K, csrc/precise.hip): the scoring GEMM is ~0.01 % of the
            # step's FLOPs and was 16 % of the bf16 logit error variance (tools/error_budget.py)
            H = trans.shape[-1]
            t3 = ops.split_bf16x3_rows(trans.detach().reshape(-1, H), rows, 0)
            w3 = ops.split_bf16x3_rows(store.m(word_name), ids, 1)
            ops.gemm_nt(t3, w3, out, bias=store.m(bias_name).index_select(0, ids.long()))
        else:
            ops.gemm_nt(transb, store.w(word_name), out, a_rows=rows, b_rows=ids, bias=store.m(bias_name), bias_by_brow=True)
        ctx.store, ctx.rows, ctx.ids, ctx.transb, ctx.shape = store, rows, ids, transb, trans.shape
        ctx.names = (word_name, bias_name)
        return out

    @staticmethod
    def backward(ctx, dlogits):
        store, rows, ids, transb = ctx.store, ctx.rows, ctx.ids, ctx.transb
        word_name, bias_name = ctx.names
        dev = dlogits.device
        R, A = dlogits.shape
        H = transb.shape[1]
        # d(rows) = dlogits @ W[ids]: M = R (a few hundred rows) x N = H with the contraction over the A scored ids -- 12 output tiles walking a
        # K = 11 k loop (0.12 ms, the longest kernel between the loss and the first backward GEMM) as one launch; run split-K as a batched product
        # (S slices of the contraction -> S x 12 tiles) + an ordered sum instead
        S = max(1, min(16, A // 768)) if R <= 1024 else 1
        Kc = ((A + 64 * S - 1) // (64 * S)) * 64
        Ap = Kc * S
        dl = torch.empty((R, Ap), device=dev, dtype=BF)
        ops.cast_pad_f32_bf16(dlogits.contiguous(), dl, R, A)
        W = store.w(word_name)
        Wg = torch.empty((A, H), device=dev, dtype=BF)
        ops.gather_rows_bf16(W, ids, Wg)
        WgT = torch.empty((H, Ap), device=dev, dtype=BF)
        ops.transpose_bf16(Wg, WgT, A, H, Ap)
        drows = torch.empty((R, H), device=dev, dtype=F32)
        if S > 1:
            parts = torch.empty((S, R, H), device=dev, dtype=F32)
            ops.gemm_nt(dl[:, :Kc], WgT[:, :Kc], parts, M=R, N=H, batch=S, stride_a=Kc, stride_b=Kc, stride_c=R * H)
            ops.sum_splits_f32(parts, drows)
        else:
            ops.gemm_nt(dl, WgT, drows)
        dtrans = torch.zeros(ctx.shape, device=dev, dtype=F32)
        ops.scatter_add_rows_f32(drows, rows, dtrans.view(-1, H))
        trows = torch.empty((R, H), device=dev, dtype=BF)
        ops.gather_rows_bf16(transb, rows, trows)
        ops.gemm_tn(dl, trows, store.g(word_name), NX=A, out_rows=ids, colsum=store.g(bias_name), colsum_by_row=True)
        return dtrans, None, None, None, None, None, None, None


_STATUS: dict = {}      # device -> int32 [1]: bit 0 set by mart_lsce_fwd (label outside [0, C) u {ignore_index}), bit 1 by mart_find_token (token absent)


def _status(dev) -> torch.Tensor:
    t = _STATUS.get(dev)
    if t is None:
        t = _STATUS[dev] = torch.zeros(1, device=dev, dtype=torch.int32)
    return t


def check_status() -> None:
    """Raise for the conditions the reference raises for at once and the stream-ordered kernels can only flag (one host sync; the trainer calls
    it where it reads the loss anyway, and after every evaluation pass):
      * a label outside [0, num_classes) that is not ``ignore_index`` reached LabelSmoothSoftmaxCEV1 (the reference's ``scatter_`` raises,
        lit_models/utils.py:54-55; here that step's loss is NaN),
      * an example without [MASK] reached the [MASK]-row lookup (``logits[arange(bs), mask_idx]`` raises a shape error in the reference,
        lit_models/transformer.py:94-95; here row 0 of that example was scored, as ``needed_rows`` was told)."""
    for dev, t in _STATUS.items():
        v = int(t.item())
        if v:
            t.zero_()
            why = []
            if v & 1:
                why.append("LabelSmoothSoftmaxCEV1: a label outside [0, num_classes) that is not ignore_index (that step's loss is NaN)")
            if v & 2:
                why.append("an example without the [MASK] token (the reference raises a shape mismatch at logits[arange(bs), mask_idx])")
            raise IndexError("; ".join(why))


check_labels = check_status


class _LSCEFn(torch.autograd.Function):
    """LabelSmoothSoftmaxCEV1.forward, lit_models/utils.py:42-66: reduction 'mean' (sum over rows / n_valid), 'sum' or 'none'; rows whose
    label == ignore_index contribute 0 and receive a zero gradient row."""

    @staticmethod
    def forward(ctx, logits, label, eps, ignore_index, reduction):
        logits = logits.contiguous()
        R = logits.shape[0]
        rows = torch.empty(R, device=logits.device, dtype=F32)
        lse = torch.empty(R, device=logits.device, dtype=F32)
        red = torch.empty(2, device=logits.device, dtype=F32) if reduction != "none" else None
        ops.lsce_fwd(logits, label, eps, rows, lse, ignore_index=ignore_index, loss_out=red, reduction=reduction, status=_status(logits.device))
        ctx.save_for_backward(logits, label, lse, red)
        ctx.eps, ctx.ignore_index, ctx.reduction = eps, ignore_index, reduction
        return rows if red is None else red[0]

    @staticmethod
    def backward(ctx, g):
        logits, label, lse, red = ctx.saved_tensors
        dl = torch.empty_like(logits)
        per_row = ctx.reduction == "none"
        ops.lsce_bwd(logits, label, lse, ctx.eps, g.contiguous().float().view(-1), 1.0, dl_f32=dl, ignore_index=ctx.ignore_index,
                     n_valid=red[1:] if ctx.reduction == "mean" else None, gscale_per_row=per_row)
        return dl, None, None, None, None


class _SimLossFn(torch.autograd.Function):
    """Relaxation loss, lit_models/transformer.py:103-108.  ``rows`` / ``L`` given: ``trans`` is the compact tensor of a row-subset pass."""

    @staticmethod
    def forward(ctx, trans, rel_idx, q_idx, a_idx, rows=None, L=None):
        trans = trans.contiguous()
        B = trans.shape[0] if rows is None else rows.shape[0]
        lrows = torch.empty(B, device=trans.device, dtype=F32)
        ops.simloss_fwd(trans, rel_idx, q_idx, a_idx, lrows, rows=rows, L_=L)
        ctx.save_for_backward(trans, rel_idx, q_idx, a_idx)
        ctx.rows, ctx.L, ctx.B = rows, L, B
        return lrows.mean()

    @staticmethod
    def backward(ctx, g):
        trans, rel_idx, q_idx, a_idx = ctx.saved_tensors
        d = torch.zeros_like(trans)
        ops.simloss_bwd(trans, rel_idx, q_idx, a_idx, g.contiguous().view(1).float(), 1.0 / ctx.B, d, rows=ctx.rows, L_=ctx.L)
        return d, None, None, None, None, None


def label_smooth_ce(logits: torch.Tensor, label: torch.Tensor, eps: float = 0.1, ignore_index: int = -100, reduction: str = "mean") -> torch.Tensor:
    if reduction not in ("mean", "sum"):
        reduction = "none"                          # lit_models/utils.py:59-62: any other string leaves the per-row losses
    return _LSCEFn.apply(logits, label.to(torch.int64).contiguous(), float(eps), int(ignore_index), reduction)


def relaxation_loss(trans: torch.Tensor, rel_idx: torch.Tensor, q_head_idx: torch.Tensor, a_head_idx: torch.Tensor) -> torch.Tensor:
    idx = (rel_idx.to(torch.int64).contiguous(), q_head_idx.to(torch.int64).contiguous(), a_head_idx.to(torch.int64).contiguous())
    rs = getattr(trans, "_mart_rows", None)
    if rs is not None:                               # the dense tensor of a row-subset pass: read (and differentiate) its compact rows
        return _SimLossFn.apply(rs.f32, *idx, rs.rows, rs.L)
    return _SimLossFn.apply(trans, *idx)


def needed_rows(input_ids: torch.Tensor, mask_token_id: int, extra=None) -> torch.Tensor:
    """The rows of ``trans_hidden_states`` a trainer-surface step reads, as ``forward(needed_rows=...)`` takes them: int32 [B, 1] (the [MASK] row) or
    [B, 5] ([MASK], rel_idx[:, 0], rel_idx[:, 1], q_head_idx, a_head_idx) FLAT ids b * L + position, built by one device launch (no host sync;
    an example without [MASK]: row 0 of that example, flagged for check_status).  The result carries ``_mart_flat`` (the model takes it as is)
    and ``_mart_mask_row`` (int32 [B]: column 0, contiguous)."""
    B, L = input_ids.shape
    dev = input_ids.device
    nr = 1 if extra is None else 5
    rows = torch.empty((B, nr), device=dev, dtype=torch.int32)
    mrow = torch.empty(B, device=dev, dtype=torch.int32)
    if extra is None:
        rel = q = a = None
    else:
        rel, q, a = (t.to(dev, torch.int64).contiguous() for t in extra)
    ops.needed_rows(input_ids.contiguous(), int(mask_token_id), rel, q, a, rows, mrow, status=_status(dev))
    rows._mart_flat, rows._mart_mask_row, rows._mart_mask_token = True, mrow, int(mask_token_id)
    return rows


def token_positions(input_ids: torch.Tensor, token_id: int) -> torch.Tensor:
    """First position of ``token_id`` in every row, int32 [B] on the device (-1: absent) -- ``(input_ids == id).nonzero()`` of
    lit_models/transformer.py:94 without the host sync."""
    pos = torch.empty(input_ids.shape[0], device=input_ids.device, dtype=torch.int32)
    ops.find_token(input_ids.contiguous(), token_id, pos, None, status=_status(input_ids.device))
    return pos


def entity_ranks(logits: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
    """1 + #(logit > logit[label]); equals argsort(argsort(-logits))[label]+1 of lit_models/transformer.py:162-164
    whenever the label's logit is not tied."""
    logits = logits.detach().contiguous().float()
    out = torch.empty(logits.shape[0], device=logits.device, dtype=torch.int64)
    ops.rank(logits, label.to(torch.int64).contiguous(), out)
    return out


class LazyRows:
    """``logits[arange(B), mask_idx]`` without the [B,L,V] tensor; supports ``[:, ids]``, ``[rows, st:ed]``, ``[rows, ids]``."""

    def __init__(self, owner: "LazyLogits", rows_i32: torch.Tensor):
        self.owner, self.rows = owner, rows_i32

    @property
    def shape(self):
        return (self.rows.numel(), self.owner.vocab)

    def _ids(self, sel) -> torch.Tensor:
        dev = self.rows.device
        if isinstance(sel, slice):
            st, ed, step = sel.indices(self.owner.vocab)
            ids = torch.arange(st, ed, step, device=dev, dtype=torch.int32)
            ids._mart_unique = True                  # a range: no uniqueness check (and no device sync) needed
            return ids
        if isinstance(sel, (list, tuple)):
            return torch.tensor(list(sel), device=dev, dtype=torch.int32)
        return sel.to(device=dev, dtype=torch.int32).contiguous()

    def __getitem__(self, key):
        rsel, csel = key if isinstance(key, tuple) else (key, slice(None))
        rows = self.rows
        if not (isinstance(rsel, slice) and rsel == slice(None)):
            rows = rows[rsel].contiguous()
        o = self.owner
        if o.compact is not None and o.precise is None:      # row-subset pass: score the compact rows (gradients go to the compact tensor)
            cidx = getattr(rows, "_mart_compact_index", None)
            if cidx is None:
                cidx = o.compact.lookup(rows)
            return _ScoreFn.apply(o.compact.f32, o.compact.bf16, cidx, self._ids(csel), o.store, o.word_name, o.bias_name, o.head_split)
        if o.precise is not None:
            if o.trans.requires_grad and hasattr(o.precise, "score_train"):
                return o.precise.score_train(o.trans, rows, self._ids(csel), o.word_name, o.bias_name)
            return o.precise.score(o.trans, rows, self._ids(csel), o.word_name, o.bias_name)
        return _ScoreFn.apply(o.trans, o.trans_bf16, rows, self._ids(csel), o.store, o.word_name, o.bias_name, o.head_split)


class LazyLogits:
    """Stand-in for ``MaskedLMOutput.logits`` [B,L,V] (2.75 GB fp32 at B=256 in the reference, modeling_unimo.py:958).
    Indexing patterns used by the trainer surface are scored on demand; ``materialize()`` builds the full tensor."""

    def __init__(self, trans: torch.Tensor, trans_bf16: torch.Tensor, store, word_name: str = WORD, bias_name: str = BIAS, precise=None,
                 head_split: bool = True, valid_rows=None, compact: Optional[RowSubset] = None):
        self.trans, self.trans_bf16, self.store = trans, trans_bf16, store
        self.compact = compact                      # RowSubset of a forward(needed_rows=...) pass: the scoring head reads the compact rows
        # forward(needed_rows=...): flat ids (b * L + position, int32 [B, n]) of the only rows of ``trans`` that were computed -- the rest is NaN
        self.valid_rows = valid_rows
        self.head_split = head_split                # the ENGINE's switch (one source of truth for the transform and the scoring GEMM)
        self.word_name, self.bias_name = word_name, bias_name
        self.precise = precise                      # engine_precise.PreciseUnimoForward: fp32-accurate scoring (eval only)
        self.vocab = store.slots[word_name].shape[0]

    @property
    def shape(self):
        return tuple(self.trans.shape[:2]) + (self.vocab,)

    def __getitem__(self, key):
        if isinstance(key, tuple) and len(key) == 2 and torch.is_tensor(key[0]) and torch.is_tensor(key[1]):
            L = self.trans.shape[1]
            pos = key[1].to(self.trans.device).to(torch.int64)
            pos = torch.where(pos < 0, pos + L, pos)
            rows = (key[0].to(self.trans.device).to(torch.int64) * L + pos).to(torch.int32).contiguous()
            if self.valid_rows is not None and not bool(torch.isin(rows, self.valid_rows.reshape(-1)).all()):     # (host sync: not the trainer's path)
                raise ValueError("logits[b, pos]: a position outside the needed_rows this forward pass was promised; call the model without "
                                 "needed_rows (or name the position) to score it")
            return LazyRows(self, rows)
        return self.materialize()[key]

    def mask_rows(self, input_ids: torch.Tensor, mask_token_id: int) -> LazyRows:
        """Device-side ``(input_ids == mask).nonzero()`` + row gather (no host sync; lit_models/transformer.py:94)."""
        B, L = input_ids.shape
        c = self.compact
        if c is not None and c.mask_index is not None and c.mask_token == int(mask_token_id):
            # the rows were built by Fn.needed_rows for this very token: slot 0 of every example is its [MASK] row -- no second search, no lookup
            row = c.mask_row
            row._mart_compact_index = c.mask_index
            return LazyRows(self, row)
        pos = torch.empty(B, device=input_ids.device, dtype=torch.int32)
        row = torch.empty(B, device=input_ids.device, dtype=torch.int32)
        ops.find_token(input_ids.contiguous(), mask_token_id, pos, row, status=_status(input_ids.device))     # absent: row b*L+0 + status bit (check_status)
        return LazyRows(self, row)

    def materialize(self) -> torch.Tensor:
        if self.valid_rows is not None:
            raise ValueError("logits.materialize(): this forward pass computed only the needed_rows it was given (every other row of "
                             "trans_hidden_states is NaN); call the model without needed_rows for the full [B, L, V] tensor")
        B, L, _ = self.trans.shape
        rows = torch.arange(B * L, device=self.trans.device, dtype=torch.int32)
        ids = torch.arange(self.vocab, device=self.trans.device, dtype=torch.int32)
        if self.precise is not None:
            return self.precise.score(self.trans, rows, ids, self.word_name, self.bias_name).view(B, L, self.vocab)
        ids._mart_unique = True
        return _ScoreFn.apply(self.trans, self.trans_bf16, rows, ids, self.store, self.word_name, self.bias_name, self.head_split).view(B, L, self.vocab)
