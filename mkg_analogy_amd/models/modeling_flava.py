"""FLAVA backbone under the MarT framework (reference: MarT/models/modeling_flava.py:2127-2204 FlavaForMaskedLM,
MarT/models/model.py:31 FlavaKGC) on the gfx950 HIP kernels -- SURVEY 8(a) row 19 / BASELINE config 4.

Only the path MarT runs is implemented: image encoder (12 pre-LN layers over 1+196+196 tokens, FlavaImageEmbeddings quirk
included), text encoder (12 pre-LN layers, FLAVA variant of the adaptive analogy reweight), the two multimodal projections
of the PRE-final-layernorm states, the 6-layer multimodal encoder, final layernorm, MLM head on the text positions.
Poolers, contrastive projections, logit_scale, codebook, ITM/MIM heads exist as parameters (checkpoint compatibility) but are
never evaluated, exactly as in the reference's MarT usage.  Sub-modules are parameter containers; compute is in
``mkg_analogy_amd.flava_engine``.  All dropout probabilities must be 0 (the facebook/flava-full config).
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, Optional

import torch
from torch import nn

from .. import functional as Fn
from .. import ops
from ..flava_engine import FlavaEngine, flava_f16_weight, flava_gemm_groups, flava_layout_order, FLAVA_DEAD
from ..params import FlatStore
from .modeling_unimo import MaskedLMOutput, _Container


class FlavaSelfAttention(_Container):
    def __init__(self, H):
        super().__init__()
        self.query, self.key, self.value = nn.Linear(H, H), nn.Linear(H, H), nn.Linear(H, H)
        self.adaptive_weight = nn.ParameterList([nn.Parameter(torch.empty(1).uniform_(0.0, 0.5)), nn.Parameter(torch.full((1,), 0.5))])


class FlavaSelfOutput(_Container):
    def __init__(self, H):
        super().__init__()
        self.dense = nn.Linear(H, H)


class FlavaAttention(_Container):
    def __init__(self, H):
        super().__init__()
        self.attention = FlavaSelfAttention(H)
        self.output = FlavaSelfOutput(H)


class _Dense(_Container):
    def __init__(self, i, o):
        super().__init__()
        self.dense = nn.Linear(i, o)


class FlavaLayer(_Container):
    def __init__(self, H, I, eps):
        super().__init__()
        self.attention = FlavaAttention(H)
        self.intermediate = _Dense(H, I)
        self.output = _Dense(I, H)
        self.layernorm_before = nn.LayerNorm(H, eps=eps)
        self.layernorm_after = nn.LayerNorm(H, eps=eps)


class FlavaEncoder(_Container):
    def __init__(self, n, H, I, eps):
        super().__init__()
        self.layer = nn.ModuleList([FlavaLayer(H, I, eps) for _ in range(n)])


class FlavaTextEmbeddings(_Container):
    def __init__(self, c):
        super().__init__()
        self.word_embeddings = nn.Embedding(c.vocab_size, c.hidden_size)
        self.position_embeddings = nn.Embedding(c.max_position_embeddings, c.hidden_size)
        self.token_type_embeddings = nn.Embedding(c.type_vocab_size, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.register_buffer("position_ids", torch.arange(c.max_position_embeddings).expand((1, -1)))


class FlavaTextModel(_Container):
    def __init__(self, c):
        super().__init__()
        self.embeddings = FlavaTextEmbeddings(c)
        self.encoder = FlavaEncoder(c.num_hidden_layers, c.hidden_size, c.intermediate_size, c.layer_norm_eps)
        self.layernorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.pooler = _Dense(c.hidden_size, c.hidden_size)


class PatchEmbeddings(_Container):
    def __init__(self, c):
        super().__init__()
        self.projection = nn.Conv2d(3, c.hidden_size, kernel_size=c.patch_size, stride=c.patch_size)


class FlavaImageEmbeddings(_Container):
    def __init__(self, c):
        super().__init__()
        self.cls_token = nn.Parameter(torch.zeros(1, 1, c.hidden_size))
        self.mask_token = nn.Parameter(torch.zeros(1, 1, c.hidden_size))
        self.patch_embeddings = PatchEmbeddings(c)
        n = (c.image_size // c.patch_size) ** 2
        self.position_embeddings = nn.Parameter(torch.zeros(1, n + 1, c.hidden_size))


class FlavaImageModel(_Container):
    def __init__(self, c):
        super().__init__()
        self.embeddings = FlavaImageEmbeddings(c)
        self.encoder = FlavaEncoder(c.num_hidden_layers, c.hidden_size, c.intermediate_size, c.layer_norm_eps)
        self.layernorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.pooler = _Dense(c.hidden_size, c.hidden_size)


class FlavaMultimodalModel(_Container):
    def __init__(self, c):
        super().__init__()
        self.cls_token = nn.Parameter(torch.zeros(1, 1, c.hidden_size))
        self.encoder = FlavaEncoder(c.num_hidden_layers, c.hidden_size, c.intermediate_size, c.layer_norm_eps)
        self.layernorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.pooler = _Dense(c.hidden_size, c.hidden_size)


class FlavaModel(_Container):
    def __init__(self, cfg):
        super().__init__()
        self.logit_scale = nn.Parameter(torch.ones([]) * 2.6592)
        self.text_model = FlavaTextModel(cfg.text_config)
        self.image_model = FlavaImageModel(cfg.image_config)
        self.multimodal_model = FlavaMultimodalModel(cfg.multimodal_config)
        H = cfg.text_config.hidden_size
        self.image_projection = nn.Linear(H, H)
        self.text_projection = nn.Linear(H, H)
        self.image_to_mm_projection = nn.Linear(H, H)
        self.text_to_mm_projection = nn.Linear(H, H)


class FlavaPredictionHeadTransform(_Container):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)


class FlavaMaskedPredictionHead(_Container):
    def __init__(self, c):
        super().__init__()
        self.transform = FlavaPredictionHeadTransform(c)
        self.decoder = nn.Linear(c.hidden_size, c.vocab_size, bias=False)
        self.bias = nn.Parameter(torch.zeros(c.vocab_size))
        self.decoder.bias = self.bias


def flava_config(vocab_size=30522, hidden_size=768, text_layers=12, image_layers=12, mm_layers=6, num_attention_heads=12,
                 intermediate_size=3072, max_position_embeddings=512, image_size=224, patch_size=16, layer_norm_eps=1e-12):
    """Plain-object stand-in for transformers.FlavaConfig (attribute access only; the real config object works too)."""
    common = dict(hidden_size=hidden_size, num_attention_heads=num_attention_heads, intermediate_size=intermediate_size,
                  layer_norm_eps=layer_norm_eps, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    return SimpleNamespace(
        text_config=SimpleNamespace(vocab_size=vocab_size, num_hidden_layers=text_layers, max_position_embeddings=max_position_embeddings,
                                    type_vocab_size=2, initializer_range=0.02, **common),
        image_config=SimpleNamespace(num_hidden_layers=image_layers, image_size=image_size, patch_size=patch_size, **common),
        multimodal_config=SimpleNamespace(num_hidden_layers=mm_layers, **common), initializer_range=0.02)


class FlavaForMaskedLM(nn.Module):
    def __init__(self, config):
        super().__init__()
        for sub in (config.text_config, config.image_config, config.multimodal_config):
            if getattr(sub, "hidden_dropout_prob", 0.0) or getattr(sub, "attention_probs_dropout_prob", 0.0):
                raise NotImplementedError("the HIP FLAVA path implements the facebook/flava-full configuration (all dropouts 0)")
        self.config = config
        self.flava = FlavaModel(config)
        self.cls = FlavaMaskedPredictionHead(config.text_config)
        self._store: Optional[FlatStore] = None
        self._engine: Optional[FlavaEngine] = None
        self.image_table = None
        self.tie_weights()

    # embedding surgery (same contract as the MKGformer class)
    def get_input_embeddings(self):
        return self.flava.text_model.embeddings.word_embeddings

    def get_output_embeddings(self):
        return self.cls.decoder

    def tie_weights(self):
        dec, emb = self.cls.decoder, self.get_input_embeddings()
        dec.weight = emb.weight
        if self.cls.bias.shape[0] != emb.weight.shape[0]:
            nb = torch.zeros(emb.weight.shape[0], dtype=self.cls.bias.dtype, device=self.cls.bias.device)
            n = min(nb.shape[0], self.cls.bias.shape[0])
            nb[:n] = self.cls.bias.data[:n]
            self.cls.bias = nn.Parameter(nb)
        dec.bias = self.cls.bias
        dec.out_features = emb.num_embeddings

    def resize_token_embeddings(self, new_num_tokens):
        old = self.get_input_embeddings()
        if new_num_tokens is None or new_num_tokens == old.weight.shape[0]:
            return
        w = old.weight.data
        new = nn.Embedding(new_num_tokens, w.shape[1]).to(device=w.device, dtype=w.dtype)
        new.weight.data.normal_(mean=0.0, std=self.config.text_config.initializer_range)
        n = min(w.shape[0], new_num_tokens)
        new.weight.data[:n] = w[:n]
        self.flava.text_model.embeddings.word_embeddings = new
        self._store = self._engine = None
        self.tie_weights()

    def finalize(self, device=None) -> FlatStore:
        if self._store is not None and self._store.still_bound():
            return self._store
        named = dict(self.named_parameters())
        if self._store is not None and self._store.owns(named):
            self._store.bind(self)
            return self._store
        ops.require_gpu()
        if device is None:
            p0 = next(iter(named.values()))
            device = p0.device if p0.is_cuda else torch.device("cuda", torch.cuda.current_device())
        c = self.config
        nl = (c.text_config.num_hidden_layers, c.image_config.num_hidden_layers, c.multimodal_config.num_hidden_layers)
        self._store = FlatStore(named, 0, torch.device(device), order=flava_layout_order(*nl), gemm_groups=flava_gemm_groups(*nl), dead=FLAVA_DEAD(nl),
                                f16_weight=flava_f16_weight)
        self.tie_weights()
        self._engine = FlavaEngine(self._store, c)
        self._anchor = torch.zeros(1, device=device, requires_grad=True)
        if not hasattr(self, "base_seed"):       # a re-finalize (resize_token_embeddings) keeps the rank-hashed seed and the step counter
            self._step = 0
            self.base_seed = 0x5EED      # dropout stream; distributed.GradSync hashes the rank into it
        if not hasattr(self, "precision"):
            self.precision = "bf16"
        self._store.bind(self)
        return self._store

    @property
    def store(self) -> FlatStore:
        return self.finalize()

    @property
    def engine(self) -> FlavaEngine:
        self.finalize()
        return self._engine

    def load_state_dict(self, state_dict, strict=True, **kw):
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        if self._store is not None and self._store.owns(dict(self.named_parameters())):
            self._store.refresh_shadows()
        return r

    def sync_shadows(self):
        self.finalize().refresh_shadows()

    def set_precision(self, precision: str):
        """"bf16" (training + eval) or "fp32" (evaluation only: fp32 activations, split-bf16 MFMA contractions, fp32 attention;
        engine_precise.PreciseFlavaForward)."""
        if precision not in ("bf16", "fp32"):
            raise ValueError("precision must be 'bf16' or 'fp32'")
        self.precision = precision
        return self

    def forward(self, input_ids=None, pixel_values=None, attention_mask=None, token_type_ids=None, bool_masked_pos=None, position_ids=None,
                image_attention_mask=None, skip_multimodal_encoder=None, output_attentions=None, output_hidden_states=True, return_dict=None,
                labels=None, sep_idx=None):
        if bool_masked_pos is not None or position_ids is not None or image_attention_mask is not None or skip_multimodal_encoder or output_attentions:
            raise NotImplementedError("MarT never passes these FlavaForMaskedLM arguments; unsupported on the HIP path")
        st = self.finalize()
        dev = st.device
        input_ids = input_ids.to(dev, torch.int64).contiguous()
        B, L = input_ids.shape
        attention_mask = torch.ones((B, L), device=dev, dtype=torch.int64) if attention_mask is None else attention_mask.to(dev, torch.int64).contiguous()
        token_type_ids = torch.zeros((B, L), device=dev, dtype=torch.int64) if token_type_ids is None else token_type_ids.to(dev, torch.int64).contiguous()
        if sep_idx is not None:
            sep_idx = sep_idx.to(dev, torch.int64).contiguous()
        pixel_values = pixel_values.to(dev, torch.float32)
        if getattr(self, "precision", "bf16") == "fp32":
            st.join_pending()
            if labels is not None:
                raise NotImplementedError("precision='fp32': no full-vocabulary labels path; score slices of .logits instead")
            if torch.is_grad_enabled():
                # verification mode: fp32-accurate forward AND backward (engine_precise.PreciseFlavaTrain); eval-mode gradients
                prt = getattr(self, "_precise_train", None)
                if prt is None or prt.st is not st:
                    from ..engine_precise import PreciseFlavaTrain
                    prt = self._precise_train = PreciseFlavaTrain(st, self.config)
                holder: Dict[str, torch.Tensor] = {}
                trans = Fn._MKGformerFn.apply(self._anchor, prt, input_ids, attention_mask, token_type_ids, pixel_values, sep_idx, False, 0, holder)
                out = MaskedLMOutput(loss=None, logits=Fn.LazyLogits(trans, None, st, word_name="flava.text_model.embeddings.word_embeddings.weight",
                                                                     bias_name="cls.bias", precise=prt), hidden_states=None, attentions=None)
                return (out, trans) if return_dict else ((out.logits,), trans)
            if self.training:
                raise NotImplementedError("precision='fp32' under no_grad is the evaluation path: call model.eval() (or set_precision('bf16'))")
            pr = getattr(self, "_precise", None)
            if pr is None or pr.st is not st:
                from ..engine_precise import PreciseFlavaForward
                pr = self._precise = PreciseFlavaForward(st, self.config)
            trans = pr.forward(input_ids, attention_mask, token_type_ids, pixel_values, sep_idx)
            out = MaskedLMOutput(loss=None, logits=Fn.LazyLogits(trans, None, st, word_name="flava.text_model.embeddings.word_embeddings.weight",
                                                                 bias_name="cls.bias", precise=pr), hidden_states=None, attentions=None)
            return (out, trans) if return_dict else ((out.logits,), trans)
        self._step += 1
        holder: Dict[str, torch.Tensor] = {}
        self._engine.save_for_backward = torch.is_grad_enabled()
        trans = Fn._MKGformerFn.apply(self._anchor, self._engine, input_ids, attention_mask, token_type_ids, pixel_values, sep_idx,
                                      bool(self.training), (self.base_seed * 1000003 + self._step * 7919) & 0x7FFFFFFFFFFF, holder)
        st.join_pending()                           # gradient zero-fill / W^T refresh issued next to this forward pass (optim.FusedAdamW)
        logits = Fn.LazyLogits(trans, holder["trans_bf16"], st, word_name="flava.text_model.embeddings.word_embeddings.weight",
                               bias_name="cls.bias", head_split=self._engine.head_split)
        loss = None
        if labels is not None:
            full = logits.materialize()
            loss = torch.nn.functional.cross_entropy(full.view(-1, full.shape[-1]), labels.to(dev).view(-1))
        out = MaskedLMOutput(loss=loss, logits=logits, hidden_states=None, attentions=None)
        if not return_dict:
            return ((loss, logits) if loss is not None else (logits,)), trans
        return out, trans
