"""MKGformer ("Unimo") behind the reference's operator API, running on the gfx950 HIP engine.

Mirrors the PUBLIC surface of MarT/models/modeling_unimo.py:839-930 (UnimoForMaskedLM): constructor
``(vision_config, text_config)``, ``forward(...)`` keyword set and 2-tuple return, ``get_input_embeddings`` /
``get_output_embeddings`` (same weight object), ``resize_token_embeddings``, ``tie_weights`` and -- for checkpoint
compatibility -- the module tree that yields the reference's 451 parameter names.

The sub-modules below are PARAMETER CONTAINERS: standard torch layers are instantiated only for their tensors and
default initialisers (the reference relies on torch defaults too, modeling_unimo.py:96-97); none of their
``forward`` methods is ever called.  All compute is in ``mkg_analogy_amd.engine`` (HIP kernels through the C ABI);
without the HIP library or a gfx950 device ``forward`` raises -- there is no eager fallback.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, Optional

import torch
from torch import nn

from .. import functional as Fn
from .. import ops
from ..engine import UnimoEngine
from ..params import FlatStore


class _Container(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container: the MKGformer forward runs in mkg_analogy_amd.engine (HIP)")


class CLIPVisionEmbeddings(_Container):
    def __init__(self, c):
        super().__init__()
        self.class_embedding = nn.Parameter(torch.randn(c.hidden_size))
        self.patch_embedding = nn.Conv2d(3, c.hidden_size, kernel_size=c.patch_size, stride=c.patch_size, bias=False)
        n = (c.image_size // c.patch_size) ** 2 + 1
        self.position_embedding = nn.Embedding(n, c.hidden_size)
        self.register_buffer("position_ids", torch.arange(n).expand((1, -1)))


class BertEmbeddings(_Container):
    def __init__(self, c):
        super().__init__()
        self.word_embeddings = nn.Embedding(c.vocab_size, c.hidden_size, padding_idx=getattr(c, "pad_token_id", 0))
        self.position_embeddings = nn.Embedding(c.max_position_embeddings, c.hidden_size)
        self.token_type_embeddings = nn.Embedding(c.type_vocab_size, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.register_buffer("position_ids", torch.arange(c.max_position_embeddings).expand((1, -1)))


class CLIPAttention(_Container):
    def __init__(self, c):
        super().__init__()
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            setattr(self, n, nn.Linear(c.hidden_size, c.hidden_size))


class CLIPMLP(_Container):
    def __init__(self, c):
        super().__init__()
        self.fc1 = nn.Linear(c.hidden_size, c.intermediate_size)
        self.fc2 = nn.Linear(c.intermediate_size, c.hidden_size)


class CLIPEncoderLayer(_Container):
    def __init__(self, c):
        super().__init__()
        self.self_attn = CLIPAttention(c)
        self.layer_norm1 = nn.LayerNorm(c.hidden_size)
        self.mlp = CLIPMLP(c)
        self.layer_norm2 = nn.LayerNorm(c.hidden_size)


class BertFusion(_Container):
    pass


class BertSelfAttention(_Container):
    def __init__(self, c):
        super().__init__()
        self.query = nn.Linear(c.hidden_size, c.hidden_size)
        self.key = nn.Linear(c.hidden_size, c.hidden_size)
        self.value = nn.Linear(c.hidden_size, c.hidden_size)
        self.fusion = BertFusion()
        # per-layer adaptive analogy weights, init U(0,.5) and 0.5 (modeling_unimo.py:305-310)
        self.adaptive_weight = nn.ParameterList([nn.Parameter(torch.empty(1).uniform_(0.0, 0.5)), nn.Parameter(torch.full((1,), 0.5))])


class BertSelfOutput(_Container):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)


class BertAttention(_Container):
    def __init__(self, c):
        super().__init__()
        self.self = BertSelfAttention(c)
        self.output = BertSelfOutput(c)


class BertIntermediate(_Container):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.intermediate_size)
        self.fusion_dense = nn.Linear(c.hidden_size, c.intermediate_size)


class BertOutput(_Container):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.intermediate_size, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)


class BertLayer(_Container):
    def __init__(self, c):
        super().__init__()
        self.attention = BertAttention(c)
        self.intermediate = BertIntermediate(c)
        self.output = BertOutput(c)


class UnimoEncoder(_Container):
    def __init__(self, vc, tc):
        super().__init__()
        self.vision_layers = nn.ModuleList([CLIPEncoderLayer(vc) for _ in range(vc.num_hidden_layers)])
        self.text_layer = nn.ModuleList([BertLayer(tc) for _ in range(tc.num_hidden_layers)])


class BertPooler(_Container):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.hidden_size)


class UnimoModel(_Container):
    def __init__(self, vc, tc, add_pooling_layer=True):
        super().__init__()
        self.vision_embeddings = CLIPVisionEmbeddings(vc)
        self.vision_pre_layrnorm = nn.LayerNorm(vc.hidden_size)
        self.vision_post_layernorm = nn.LayerNorm(vc.hidden_size)      # present in checkpoints, never used (:683)
        self.text_embeddings = BertEmbeddings(tc)
        self.text_pooler = BertPooler(tc) if add_pooling_layer else None  # dead compute in the reference (:748)
        self.encoder = UnimoEncoder(vc, tc)


class BertPredictionHeadTransform(_Container):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)


class UnimoLMPredictionHead(_Container):
    def __init__(self, c):
        super().__init__()
        self.transform = BertPredictionHeadTransform(c)
        self.decoder = nn.Linear(c.hidden_size, c.vocab_size, bias=False)
        self.bias = nn.Parameter(torch.zeros(c.vocab_size))
        self.decoder.bias = self.bias


class UnimoOnlyMLMHead(_Container):
    def __init__(self, c):
        super().__init__()
        self.predictions = UnimoLMPredictionHead(c)


class MaskedLMOutput(SimpleNamespace):
    """Duck-type of transformers.modeling_outputs.MaskedLMOutput: ``out.logits`` and ``out[0]`` style access."""

    def __getitem__(self, i):
        return [v for v in (self.loss, self.logits, self.hidden_states, self.attentions) if v is not None][i]


class UnimoForMaskedLM(nn.Module):
    accepts_needed_rows = True                      # forward(needed_rows=...): last-layer row subset

    def __init__(self, vision_config, text_config):
        super().__init__()
        self.unimo = UnimoModel(vision_config, text_config)
        self.cls = UnimoOnlyMLMHead(text_config)
        self.config = text_config
        self.vision_config = vision_config
        self._store: Optional[FlatStore] = None
        self._engine: Optional[UnimoEngine] = None
        self._step = 0
        self.image_table = None                     # optional resident [N_img,3,S,S] f32 table for device-side batch assembly
        self.base_seed = 0x5EED
        self.precision = "bf16"                     # "fp32": fp32-accurate evaluation path (engine_precise), forward only
        self._precise = None
        self._precise_train = None
        self._mask_index: dict = {}                  # (device, B, nr) -> compact indices of slot 0 (row-subset passes)
        self.tie_weights()

    # ------------------------------------------------------------------ embedding surgery (modeling_unimo.py:895-930)
    def get_input_embeddings(self):
        return self.unimo.text_embeddings.word_embeddings

    def get_output_embeddings(self):
        return self.cls.predictions.decoder

    def set_output_embeddings(self, new_embeddings):
        self.cls.predictions.decoder = new_embeddings

    def tie_weights(self):
        dec, emb = self.get_output_embeddings(), self.get_input_embeddings()
        dec.weight = emb.weight
        pred = self.cls.predictions
        if pred.bias.shape[0] != emb.weight.shape[0]:                   # zero-pad / truncate the decoder bias on resize
            nb = torch.zeros(emb.weight.shape[0], dtype=pred.bias.dtype, device=pred.bias.device)
            n = min(nb.shape[0], pred.bias.shape[0])
            nb[:n] = pred.bias.data[:n]
            pred.bias = nn.Parameter(nb)
        dec.bias = pred.bias
        dec.out_features = emb.num_embeddings

    def resize_token_embeddings(self, new_num_tokens):
        old = self.get_input_embeddings()
        if new_num_tokens is None or new_num_tokens == old.weight.shape[0]:
            return
        w = old.weight.data
        new = nn.Embedding(new_num_tokens, w.shape[1]).to(device=w.device, dtype=w.dtype)
        new.weight.data.normal_(mean=0.0, std=self.config.initializer_range)   # _init_text_weights, :768-771
        n = min(w.shape[0], new_num_tokens)
        new.weight.data[:n] = w[:n]
        self.unimo.text_embeddings.word_embeddings = new
        self._store = self._engine = None
        self.tie_weights()

    # ------------------------------------------------------------------ flat storage / engine
    def _named(self) -> Dict[str, nn.Parameter]:
        return dict(self.named_parameters())          # tied decoder weight/bias are deduplicated by torch

    def finalize(self, device=None) -> FlatStore:
        """Move the parameters into the flat fp32/bf16 buffers the kernels use (idempotent)."""
        if self._store is not None and self._store.still_bound():
            return self._store
        named = self._named()
        if self._store is not None and self._store.owns(named):
            self._store.bind(self)
            return self._store
        ops.require_gpu()
        if device is None:
            p0 = next(iter(named.values()))
            device = p0.device if p0.is_cuda else torch.device("cuda", torch.cuda.current_device())
        self._store = FlatStore(named, self.config.num_hidden_layers, torch.device(device))
        self.tie_weights()
        for b in ("unimo.vision_embeddings.position_ids", "unimo.text_embeddings.position_ids"):
            mod, _, name = b.rpartition(".")
            m = self.get_submodule(mod)
            m._buffers[name] = m._buffers[name].to(device)
        self._engine = UnimoEngine(self._store, self.vision_config, self.config)
        self._anchor = torch.zeros(1, device=device, requires_grad=True)
        self._store.bind(self)
        return self._store

    @property
    def store(self) -> FlatStore:
        return self.finalize()

    @property
    def engine(self) -> UnimoEngine:
        self.finalize()
        return self._engine

    def load_state_dict(self, state_dict, strict=True, **kw):
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        if self._store is not None and self._store.owns(self._named()):
            self._store.refresh_shadows()
        return r

    def set_precision(self, precision: str):
        """"bf16" (default: bf16 MFMA operands, fp32 accumulation / residual streams; training + eval) or "fp32"
        (fp32 activations, split-bf16 MFMA contractions, fp32 attention -- meets the reference's fp32 results to ~1e-4 on
        logits; evaluation under no_grad, and a verification-mode training step (eval-mode gradients, no dropout replay) when
        gradients are enabled; see engine_precise.py)."""
        if precision not in ("bf16", "fp32"):
            raise ValueError("precision must be 'bf16' or 'fp32'")
        self.precision = precision
        return self

    def set_image_table(self, table: torch.Tensor):
        """Keep the per-entity pixel tensor (MarT/tools/encode_images_data.py output, data_module.py:209) resident in HBM so
        that batches carry [B,2] row indices instead of 1.2 MB of pixels per example."""
        st = self.finalize()
        self.image_table = table.to(st.device, torch.float32).contiguous()

    def sync_shadows(self):
        """Call after editing parameters in place (e.g. _init_relation_word) so the bf16 GEMM operands follow."""
        self.finalize().refresh_shadows()

    # ------------------------------------------------------------------ forward (modeling_unimo.py:848-893)
    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, head_mask=None, sep_idx=None,
                pixel_values=None, output_attentions=None, output_hidden_states=None, return_dict=None, labels=None, image_index=None,
                needed_rows=None):
        """``needed_rows`` (optional, int [B] or [B, n] token positions): a promise that only these rows of ``trans_hidden_states`` / ``logits`` will
        be read (the trainer surface reads the [MASK] row and four more, lit_models/transformer.py:94-95,103-107).  The last text layer's
        post-attention part and the head transform then run on those rows only; the returned ``trans_hidden_states`` is NaN elsewhere, and
        ``logits`` refuses to score (or materialise) any other row.
        Exact for the rows named; omit it to get every row as the reference does."""
        if output_attentions:
            raise NotImplementedError("attention maps are never materialised by the flash-style attention kernels (output_attentions is unused by MarT)")
        if output_hidden_states and self.precision != "bf16":
            raise NotImplementedError("output_hidden_states is served by the bf16 engine's stream taps; call set_precision('bf16') for it")
        if position_ids is not None or head_mask is not None:
            raise NotImplementedError("position_ids / head_mask are unused by MarT (modeling_unimo.py:78-80) and unsupported here")
        st = self.finalize()
        dev = st.device
        input_ids = input_ids.to(dev, torch.int64).contiguous()
        B, L = input_ids.shape
        attention_mask = torch.ones((B, L), device=dev, dtype=torch.int64) if attention_mask is None else attention_mask.to(dev, torch.int64).contiguous()
        token_type_ids = torch.zeros((B, L), device=dev, dtype=torch.int64) if token_type_ids is None else token_type_ids.to(dev, torch.int64).contiguous()
        if sep_idx is not None:
            sep_idx = sep_idx.to(dev, torch.int64).contiguous()
        image_table = None
        if image_index is not None:                 # device-side batch assembly (mkg_analogy_amd.batching.DeviceImageTable)
            if self.image_table is None:
                raise ValueError("image_index given but no image table attached: call model.set_image_table(table) first")
            image_table = self.image_table
            image_index = image_index.to(dev, torch.int32).contiguous()
        else:
            pixel_values = pixel_values.to(dev, torch.float32)
        train = bool(self.training)
        if self.precision == "fp32":
            st.join_pending()                       # (the fp32-accurate engines read W^T / write gradients from their first kernels on)
            if labels is not None:
                raise NotImplementedError("precision='fp32': no full-vocabulary labels path; score slices of .logits instead")
            if torch.is_grad_enabled():
                # verification mode: fp32-accurate forward AND backward (engine_precise.PreciseUnimoTrain); eval-mode gradients only
                if self._precise_train is None or self._precise_train.st is not st:
                    from ..engine_precise import PreciseUnimoTrain
                    self._precise_train = PreciseUnimoTrain(st, self.vision_config, self.config)
                holder: Dict[str, torch.Tensor] = {}
                trans = Fn._MKGformerFn.apply(self._anchor, self._precise_train, input_ids, attention_mask, token_type_ids, pixel_values, sep_idx,
                                              train, 0, holder, image_table, image_index)
                out = MaskedLMOutput(loss=None, logits=Fn.LazyLogits(trans, None, st, precise=self._precise_train), hidden_states=None, attentions=None)
                return (out, trans) if return_dict else ((out.logits,), trans)
            if train:
                raise NotImplementedError("precision='fp32' under no_grad is the evaluation path: call model.eval() (or set_precision('bf16'))")
            if self._precise is None or self._precise.st is not st:
                from ..engine_precise import PreciseUnimoForward
                self._precise = PreciseUnimoForward(st, self.vision_config, self.config)
            trans = self._precise.forward(input_ids, attention_mask, token_type_ids, pixel_values, sep_idx, image_table=image_table,
                                          image_index=image_index)
            out = MaskedLMOutput(loss=None, logits=Fn.LazyLogits(trans, None, st, precise=self._precise), hidden_states=None, attentions=None)
            return (out, trans) if return_dict else ((out.logits,), trans)
        self._step += 1
        seed = (self.base_seed * 1000003 + self._step * 7919) & 0x7FFFFFFFFFFF
        holder: Dict[str, torch.Tensor] = {}
        self._engine.save_for_backward = torch.is_grad_enabled()
        rows = mask_row = None
        if output_hidden_states:
            # modeling_unimo.py:604-646: the text stream entering every layer + the last layer's output (13 tensors [B, L, H]).  Served by the engine's
            # stream taps: f32 copies of its text stream, DETACHED (no gradient flows through them); the pass runs dense (a row subset has no layer-11 rows)
            needed_rows = None
            self._engine.taps = {}
        if needed_rows is not None and labels is None:
            if getattr(needed_rows, "_mart_flat", False):               # built by Fn.needed_rows: flat int32 ids already (one device launch)
                rows, mask_row = needed_rows, getattr(needed_rows, "_mart_mask_row", None)
                assert rows.dtype == torch.int32 and rows.dim() == 2 and rows.shape[0] == B and rows.device == dev
            else:
                nr_ = needed_rows.to(dev).reshape(B, -1).to(torch.int64)
                nr_ = torch.where(nr_ < 0, nr_ + L, nr_).clamp(0, L - 1)     # negative positions wrap as in the reference's fancy indexing; out of place
                rows = (torch.arange(B, device=dev, dtype=torch.int64)[:, None] * L + nr_).to(torch.int32).contiguous()
        hidden_states = None
        try:
            trans = Fn._MKGformerFn.apply(self._anchor, self._engine, input_ids, attention_mask, token_type_ids, pixel_values, sep_idx, train, seed, holder,
                                          image_table, image_index, rows)
            if output_hidden_states:
                tp = self._engine.taps
                hidden_states = (tp["txt_emb"],) + tuple(tp[f"txt{l}"] for l in range(self.config.num_hidden_layers))
        finally:
            if output_hidden_states:
                self._engine.taps = None
        st.join_pending()                           # gradient zero-fill / W^T refresh issued next to this forward pass (optim.FusedAdamW)
        compact = None
        if rows is not None and trans.dim() == 2:   # row-subset pass: the engine returned the compact [B * nr, H] rows
            nr = int(rows.shape[1])
            mi = None
            if mask_row is not None:
                key = (dev, B, nr)
                mi = self._mask_index.get(key)
                if mi is None:
                    mi = self._mask_index[key] = (torch.arange(B, device=dev, dtype=torch.int32) * nr).contiguous()
            compact = Fn.RowSubset(trans, holder["trans_bf16"], rows, L, mask_index=mi, mask_row=mask_row,
                                   mask_token=getattr(needed_rows, "_mart_mask_token", None))
            trans = Fn._DenseRowsFn.apply(trans, rows, B, L)            # the reference's [B, L, H] tensor: promised rows filled in, NaN elsewhere
            trans._mart_rows = compact
        logits = Fn.LazyLogits(trans, holder["trans_bf16"] if compact is None else None, st, head_split=self._engine.head_split, valid_rows=rows,
                               compact=compact)
        loss = None
        if labels is not None:                      # CrossEntropyLoss over the full vocabulary (:880-882); not used by MarT
            full = logits.materialize()
            loss = torch.nn.functional.cross_entropy(full.view(-1, full.shape[-1]), labels.to(dev).view(-1))
        out = MaskedLMOutput(loss=loss, logits=logits, hidden_states=hidden_states, attentions=None)
        if not return_dict:
            return tuple(v for v in (loss, logits, hidden_states) if v is not None), trans
        return out, trans
