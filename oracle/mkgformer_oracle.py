"""CPU oracle for the MarT / MKGformer analogy hot path.

TEST INFRASTRUCTURE ONLY.  This file is a plain-PyTorch fp32 CPU restatement of
the reference algorithm.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product path
(``mkg_analogy_amd``) never does and fails loudly without its HIP extension.

Parity status: PINNED against the reference itself.  ``oracle/gen_goldens.py``
imports the unmodified reference from ``/root/reference`` in the build
container, runs it on seeded inputs and stores inputs+outputs under
``tests/golden``; ``tests/test_oracle_vs_golden.py`` checks this file against
those vectors (the reference has no tests / golden vectors of its own,
SURVEY.md section 4).

It is written functionally over a ``{state_dict_name: tensor}`` mapping that
uses the reference's 451 parameter names, so the same weights drive the
reference, this oracle and the HIP product.

Each function cites the reference lines (relative to /root/reference) it restates.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------- configs
@dataclass
class TextCfg:
    """bert-base-uncased defaults (MarT/main.py:80)."""
    vocab_size: int = 30522
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    max_position_embeddings: int = 512
    type_vocab_size: int = 2
    layer_norm_eps: float = 1e-12
    hidden_dropout_prob: float = 0.1
    attention_probs_dropout_prob: float = 0.1
    initializer_range: float = 0.02


@dataclass
class VisionCfg:
    """clip-vit-base-patch32 vision defaults (MarT/main.py:79); patch_size=16 -> P=196."""
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    image_size: int = 224
    patch_size: int = 32
    attention_dropout: float = 0.0
    # modeling_unimo.py:486-488,682 use nn.LayerNorm defaults, i.e. 1e-5, ignoring the config
    layer_norm_eps: float = 1e-5

    @property
    def num_patches(self) -> int:
        return (self.image_size // self.patch_size) ** 2


# --------------------------------------------------------------------------- parameter table
def param_shapes(vc: VisionCfg, tc: TextCfg) -> Dict[str, Tuple[int, ...]]:
    """Names and shapes of the reference's trainable tensors, in ``named_parameters()`` order.

    Follows the module construction order of modeling_unimo.py:676-693 (UnimoModel),
    :580-587 (UnimoEncoder), :839-846 / :942-954 (head).  The decoder weight is tied to
    the word embedding (:904-913) and therefore not listed separately.
    """
    H, I, Hv, Iv = tc.hidden_size, tc.intermediate_size, vc.hidden_size, vc.intermediate_size
    s: Dict[str, Tuple[int, ...]] = {}
    u = "unimo."
    s[u + "vision_embeddings.class_embedding"] = (Hv,)
    s[u + "vision_embeddings.patch_embedding.weight"] = (Hv, 3, vc.patch_size, vc.patch_size)
    s[u + "vision_embeddings.position_embedding.weight"] = (vc.num_patches + 1, Hv)
    for ln in ("vision_pre_layrnorm", "vision_post_layernorm"):
        s[u + ln + ".weight"] = (Hv,)
        s[u + ln + ".bias"] = (Hv,)
    s[u + "text_embeddings.word_embeddings.weight"] = (tc.vocab_size, H)
    s[u + "text_embeddings.position_embeddings.weight"] = (tc.max_position_embeddings, H)
    s[u + "text_embeddings.token_type_embeddings.weight"] = (tc.type_vocab_size, H)
    s[u + "text_embeddings.LayerNorm.weight"] = (H,)
    s[u + "text_embeddings.LayerNorm.bias"] = (H,)
    s[u + "text_pooler.dense.weight"] = (H, H)
    s[u + "text_pooler.dense.bias"] = (H,)
    for i in range(vc.num_hidden_layers):
        p = f"{u}encoder.vision_layers.{i}."
        for proj in ("k_proj", "v_proj", "q_proj", "out_proj"):
            s[p + f"self_attn.{proj}.weight"] = (Hv, Hv)
            s[p + f"self_attn.{proj}.bias"] = (Hv,)
        s[p + "layer_norm1.weight"] = (Hv,)
        s[p + "layer_norm1.bias"] = (Hv,)
        s[p + "mlp.fc1.weight"] = (Iv, Hv)
        s[p + "mlp.fc1.bias"] = (Iv,)
        s[p + "mlp.fc2.weight"] = (Hv, Iv)
        s[p + "mlp.fc2.bias"] = (Hv,)
        s[p + "layer_norm2.weight"] = (Hv,)
        s[p + "layer_norm2.bias"] = (Hv,)
    for i in range(tc.num_hidden_layers):
        p = f"{u}encoder.text_layer.{i}."
        for proj in ("query", "key", "value"):
            s[p + f"attention.self.{proj}.weight"] = (H, H)
            s[p + f"attention.self.{proj}.bias"] = (H,)
        s[p + "attention.self.adaptive_weight.0"] = (1,)
        s[p + "attention.self.adaptive_weight.1"] = (1,)
        s[p + "attention.output.dense.weight"] = (H, H)
        s[p + "attention.output.dense.bias"] = (H,)
        s[p + "attention.output.LayerNorm.weight"] = (H,)
        s[p + "attention.output.LayerNorm.bias"] = (H,)
        s[p + "intermediate.dense.weight"] = (I, H)
        s[p + "intermediate.dense.bias"] = (I,)
        s[p + "intermediate.fusion_dense.weight"] = (I, H)
        s[p + "intermediate.fusion_dense.bias"] = (I,)
        s[p + "output.dense.weight"] = (H, I)
        s[p + "output.dense.bias"] = (H,)
        s[p + "output.LayerNorm.weight"] = (H,)
        s[p + "output.LayerNorm.bias"] = (H,)
    s["cls.predictions.bias"] = (tc.vocab_size,)
    s["cls.predictions.transform.dense.weight"] = (H, H)
    s["cls.predictions.transform.dense.bias"] = (H,)
    s["cls.predictions.transform.LayerNorm.weight"] = (H,)
    s["cls.predictions.transform.LayerNorm.bias"] = (H,)
    return s


def init_params(vc: VisionCfg, tc: TextCfg, seed: int = 0, dtype=torch.float32) -> Dict[str, Tensor]:
    """Synthetic weights per SURVEY 8(d): N(0,0.02) matrices/embeddings, LN (1,0),
    zero biases, adaptive_weight=(0.25,0.5).  Generated with numpy PCG64 so every side
    (reference import, oracle, product) can regenerate the same values from the seed."""
    rng = np.random.default_rng(seed)
    out: Dict[str, Tensor] = {}
    for name, shape in param_shapes(vc, tc).items():
        if name.endswith("adaptive_weight.0"):
            v = np.full(shape, 0.25, np.float32)
        elif name.endswith("adaptive_weight.1"):
            v = np.full(shape, 0.5, np.float32)
        elif ("LayerNorm.weight" in name or "layer_norm1.weight" in name or "layer_norm2.weight" in name
              or "layrnorm.weight" in name or "layernorm.weight" in name):
            v = (1.0 + 0.05 * rng.standard_normal(shape)).astype(np.float32)
        elif name.endswith(".bias") or name == "cls.predictions.bias":
            v = (0.02 * rng.standard_normal(shape)).astype(np.float32)
        else:
            v = (0.02 * rng.standard_normal(shape)).astype(np.float32)
        out[name] = torch.from_numpy(v).to(dtype)
    return out


def condition_weights(sd: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """Well-conditioned variant of the synthetic weights (test infrastructure; shared by the golden generators and the tests).
    With plain N(0,0.02) weights the UNSCALED fusion softmax softmax(ctx vis^T) of text layers 8-11 (modeling_unimo.py:405-410)
    is nearly one-hot (score std ~15) and the network is chaotic there: rounding only the weight matrices to bf16 inside the
    fp32 math already moves trans_hidden by ~5 %.  Shrinking the text value projection of those layers keeps every code path
    live and makes the map smooth, so bf16 implementations can be held to the 1e-2 logit tolerance.  Returns a new dict."""
    out = dict(sd)
    for l in range(8, 12):
        for k in ("weight", "bias"):
            n = f"unimo.encoder.text_layer.{l}.attention.self.value.{k}"
            out[n] = sd[n] * 0.05
    return out


def outlier_weights(sd: Dict[str, Tensor], channels=(5, 77, 300, 511, 700), gain: float = 12.0, offset: float = 2.0,
                    residual_gain: float = 2.0, fusion_comp: bool = True) -> Dict[str, Tensor]:
    """Synthetic weights with the activation statistics trained CLIP / BERT checkpoints are known for (none can be loaded offline; test
    infrastructure): a handful of OUTLIER CHANNELS -- every LayerNorm scales them by ``gain`` and shifts them by ``offset``, so post-LayerNorm
    activations carry entries ~10-30 where the rest is O(1), with large per-row means -- and LARGE RESIDUAL NORMS (the vision out-proj / fc2 and the
    text attention-output / output projections are scaled by ``residual_gain``, so the f32 residual streams grow from layer to layer instead of
    staying at the embedding scale).  Applied on top of ``condition_weights``; ``fusion_comp`` keeps the score scale of the unscaled fusion softmax
    (which reads the vision residual stream directly) where ``condition_weights`` put it, so the map stays smooth: the reference's own bf16-weight control
    moves the logits by ~1e-2 on this set (3e-3 .. 1.5e-2 for residual_gain 1 .. 3), as on the conditioned set without outliers.  Returns a new dict."""
    out = condition_weights(sd)
    ch = torch.as_tensor(list(channels), dtype=torch.long)
    for k, v in list(out.items()):
        if v.dim() == 1 and ("LayerNorm.weight" in k or "layer_norm1.weight" in k or "layer_norm2.weight" in k or "layrnorm.weight" in k):
            w = v.clone(); w[ch] = w[ch] * gain; out[k] = w
        elif v.dim() == 1 and ("LayerNorm.bias" in k or "layer_norm1.bias" in k or "layer_norm2.bias" in k or "layrnorm.bias" in k):
            b = v.clone(); b[ch] = b[ch] + offset; out[k] = b
        elif v.dim() == 2 and (k.endswith("self_attn.out_proj.weight") or k.endswith("mlp.fc2.weight") or
                               (("text_layer" in k) and (k.endswith("attention.output.dense.weight") or k.endswith(".output.dense.weight")))):
            out[k] = v * residual_gain
    if fusion_comp:
        # the unscaled fusion softmax softmax(ctx vis^T) sees the vision residual stream directly: keep its score scale where condition_weights put it
        for l in range(8, 12):
            for k in ("weight", "bias"):
                n = f"unimo.encoder.text_layer.{l}.attention.self.value.{k}"
                out[n] = out[n] / (residual_gain * residual_gain)
    return out


# --------------------------------------------------------------------------- small pieces
def gelu_erf(x: Tensor) -> Tensor:
    """transformers ACT2FN['gelu'] = exact erf GELU (call sites modeling_unimo.py:454,967)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def quick_gelu(x: Tensor) -> Tensor:
    """transformers ACT2FN['quick_gelu'] = x*sigmoid(1.702x) (call site modeling_unimo.py:279)."""
    return x * torch.sigmoid(1.702 * x)


def _ln(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def _lin(x: Tensor, sd: Dict[str, Tensor], prefix: str) -> Tensor:
    return F.linear(x, sd[prefix + ".weight"], sd[prefix + ".bias"])


def extended_mask(attention_mask: Tensor) -> Tensor:
    """modeling_unimo.py:44,55-56: (1 - mask.long()) * -10000.0, shape [B,1,1,L]."""
    return (1.0 - attention_mask[:, None, None, :].to(torch.long)) * -10000.0


def _split_heads(x: Tensor, nh: int) -> Tensor:
    B, S, H = x.shape
    return x.view(B, S, nh, H // nh).permute(0, 2, 1, 3)


def _merge_heads(x: Tensor) -> Tensor:
    B, nh, S, dh = x.shape
    return x.permute(0, 2, 1, 3).reshape(B, S, nh * dh)


# --------------------------------------------------------------------------- embeddings
def vision_embed(sd, vc: VisionCfg, pixel_values: Tensor) -> Tensor:
    """CLIPVisionEmbeddings.forward (modeling_unimo.py:119-132) + vision_pre_layrnorm (:711).

    pixel_values [B,2,3,S,S].  Token order [cls, img0 patches (row-major), img1 patches];
    position ids [0,1..P,1..P].
    """
    u = "unimo.vision_embeddings."
    w = sd[u + "patch_embedding.weight"]
    B = pixel_values.shape[0]
    p = vc.patch_size
    patches = []
    for img in range(2):
        e = F.conv2d(pixel_values[:, img], w, None, stride=p)       # [B,H,g,g]
        patches.append(e.flatten(2).transpose(1, 2))                # [B,P,H]
    cls = sd[u + "class_embedding"].expand(B, 1, -1)
    x = torch.cat([cls] + patches, dim=1)
    pos = sd[u + "position_embedding.weight"]
    x = x + torch.cat([pos, pos[1:]], dim=0)[None]
    return _ln(x, sd["unimo.vision_pre_layrnorm.weight"], sd["unimo.vision_pre_layrnorm.bias"], vc.layer_norm_eps)


def text_embed(sd, tc: TextCfg, input_ids: Tensor, token_type_ids: Tensor, train: bool) -> Tensor:
    """BertEmbeddings.forward (modeling_unimo.py:152-186)."""
    u = "unimo.text_embeddings."
    L = input_ids.shape[1]
    x = sd[u + "word_embeddings.weight"][input_ids] + sd[u + "token_type_embeddings.weight"][token_type_ids]
    x = x + sd[u + "position_embeddings.weight"][:L][None]
    x = _ln(x, sd[u + "LayerNorm.weight"], sd[u + "LayerNorm.bias"], tc.layer_norm_eps)
    return F.dropout(x, tc.hidden_dropout_prob, train)


# --------------------------------------------------------------------------- layers
def vision_layer(sd, vc: VisionCfg, idx: int, x: Tensor, prefix_kv: Optional[Tuple[Tensor, Tensor]]) -> Tensor:
    """CLIPEncoderLayer.forward (modeling_unimo.py:490-527) with CLIPAttention (:212-272)
    and CLIPMLP (:283-287).  prefix_kv = text (K,V) [B,nh,L,dh] prepended on the key axis."""
    p = f"unimo.encoder.vision_layers.{idx}."
    nh = vc.num_attention_heads
    dh = vc.hidden_size // nh
    h = _ln(x, sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], vc.layer_norm_eps)
    q = _split_heads(_lin(h, sd, p + "self_attn.q_proj") * (dh ** -0.5), nh)
    k = _split_heads(_lin(h, sd, p + "self_attn.k_proj"), nh)
    v = _split_heads(_lin(h, sd, p + "self_attn.v_proj"), nh)
    if prefix_kv is not None:
        k = torch.cat([prefix_kv[0], k], dim=2)
        v = torch.cat([prefix_kv[1], v], dim=2)
    a = torch.softmax(q @ k.transpose(-1, -2), dim=-1)              # no mask, attention_dropout = 0
    ctx = _merge_heads(a @ v)
    x = x + _lin(ctx, sd, p + "self_attn.out_proj")
    h = _ln(x, sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], vc.layer_norm_eps)
    h = _lin(quick_gelu(_lin(h, sd, p + "mlp.fc1")), sd, p + "mlp.fc2")
    return x + h


def reweight_factor(w0: Tensor, w1: Tensor, sep: Tensor, L: int) -> Tensor:
    """The multiplicative form of the in-place adaptive analogy reweight (modeling_unimo.py:342-349):
    columns >= s are scaled by clamp(w0,0,.5) on rows < s and by clamp(w1,.5,1) on rows >= s,
    s = sep_idx[i,2]; columns < s keep factor 1.  Returns [B,1,L,L]."""
    c0 = torch.clamp(w0, 0, 0.5)
    c1 = torch.clamp(w1, 0.5, 1)
    ar = torch.arange(L, device=sep.device)
    s = sep[:, None]
    col_hi = (ar[None, :] >= s)[:, None, :]                          # [B,1,L] columns >= s
    row_lo = (ar[None, :] < s)[:, :, None]                           # [B,L,1] rows < s
    one = torch.ones((), dtype=c0.dtype, device=sep.device)
    f = torch.where(col_hi, torch.where(row_lo, c0, c1), one)
    return f[:, None]


def fusion(ctx: Tensor, vis: Tensor) -> Tensor:
    """BertFusion.forward (modeling_unimo.py:400-414): softmax(ctx vis^T) vis, unscaled, unmasked."""
    return torch.softmax(ctx @ vis.transpose(-1, -2), dim=-1) @ vis


def text_layer(sd, tc: TextCfg, idx: int, x: Tensor, ext_mask: Tensor, sep_idx: Optional[Tensor],
               vis: Optional[Tensor], train: bool):
    """BertLayer.forward (modeling_unimo.py:540-577): BertSelfAttention (:317-377),
    BertSelfOutput (:387-391), BertIntermediate (:458-464), BertOutput (:474-478).
    Returns (layer_output, (K,V)) with K,V [B,nh,L,dh] as exported at :336."""
    p = f"unimo.encoder.text_layer.{idx}."
    nh = tc.num_attention_heads
    dh = tc.hidden_size // nh
    L = x.shape[1]
    q = _split_heads(_lin(x, sd, p + "attention.self.query"), nh)
    k = _split_heads(_lin(x, sd, p + "attention.self.key"), nh)
    v = _split_heads(_lin(x, sd, p + "attention.self.value"), nh)
    scores = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    if sep_idx is not None:
        scores = scores * reweight_factor(sd[p + "attention.self.adaptive_weight.0"],
                                          sd[p + "attention.self.adaptive_weight.1"], sep_idx[:, 2], L)
    scores = scores + ext_mask
    probs = F.dropout(torch.softmax(scores, dim=-1), tc.attention_probs_dropout_prob, train)
    ctx = _merge_heads(probs @ v)
    fus = fusion(ctx, vis) if vis is not None else None
    a = _lin(ctx, sd, p + "attention.output.dense")
    a = F.dropout(a, tc.hidden_dropout_prob, train)
    a = _ln(a + x, sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"], tc.layer_norm_eps)
    h = _lin(a, sd, p + "intermediate.dense")
    if fus is not None:
        h = h + _lin(fus, sd, p + "intermediate.fusion_dense")
    h = gelu_erf(h)
    o = F.dropout(_lin(h, sd, p + "output.dense"), tc.hidden_dropout_prob, train)
    o = _ln(o + a, sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], tc.layer_norm_eps)
    return o, (k, v)


def encoder(sd, vc: VisionCfg, tc: TextCfg, vis: Tensor, txt: Tensor, ext_mask: Tensor,
            sep_idx: Optional[Tensor], train: bool, taps: Optional[dict] = None) -> Tensor:
    """UnimoEncoder.forward cross-wiring (modeling_unimo.py:589-658): vision layer idx takes the
    (K,V) of text layer idx-1 iff idx>=8 (:616); text layer idx fuses with the output of vision
    layer idx iff idx>=8 (:627); text exports (K,V) iff idx>=7 (:628)."""
    kv = None
    for idx in range(vc.num_hidden_layers):
        vis = vision_layer(sd, vc, idx, vis, kv if idx >= 8 else None)
        txt, kv_new = text_layer(sd, tc, idx, txt, ext_mask, sep_idx, vis if idx >= 8 else None, train)
        kv = kv_new if idx >= 7 else None
        if taps is not None:
            taps[f"vis{idx}"] = vis
            taps[f"txt{idx}"] = txt
    return txt


def head_transform(sd, tc: TextCfg, seq: Tensor) -> Tensor:
    """BertPredictionHeadTransform.forward (modeling_unimo.py:972-975)."""
    p = "cls.predictions.transform."
    return _ln(gelu_erf(_lin(seq, sd, p + "dense")), sd[p + "LayerNorm.weight"], sd[p + "LayerNorm.bias"], tc.layer_norm_eps)


def forward(sd, vc: VisionCfg, tc: TextCfg, input_ids, attention_mask, token_type_ids, pixel_values,
            sep_idx=None, train: bool = False, full_logits: bool = False, taps: Optional[dict] = None):
    """UnimoForMaskedLM.forward (modeling_unimo.py:848-893) -> (logits or None, trans_hidden [B,L,H]).

    ``full_logits=True`` materialises the reference's [B,L,V] tensor (:958); otherwise only
    trans_hidden is returned and callers score the rows/columns they need with ``score``.
    The dead text_pooler (:748) is not evaluated.
    """
    vis = vision_embed(sd, vc, pixel_values)
    txt = text_embed(sd, tc, input_ids, token_type_ids, train)
    seq = encoder(sd, vc, tc, vis, txt, extended_mask(attention_mask), sep_idx, train, taps)
    trans = head_transform(sd, tc, seq)
    logits = None
    if full_logits:
        logits = F.linear(trans, sd["unimo.text_embeddings.word_embeddings.weight"], sd["cls.predictions.bias"])
    return logits, trans


def score(sd, trans_rows: Tensor, ids) -> Tensor:
    """Columns ``ids`` of the tied decoder (modeling_unimo.py:904-913,958) applied to [B,H] rows:
    equals ``logits[rows][:, ids]`` of the reference (lit_models/transformer.py:95)."""
    W = sd["unimo.text_embeddings.word_embeddings.weight"]
    b = sd["cls.predictions.bias"]
    if isinstance(ids, slice):
        return F.linear(trans_rows, W[ids], b[ids])
    ids = torch.as_tensor(ids, dtype=torch.long)
    return F.linear(trans_rows, W[ids], b[ids])


# --------------------------------------------------------------------------- losses / ranking
def label_smooth_ce(logits: Tensor, label: Tensor, eps: float = 0.1, ignore_index: int = -100, reduction: str = "mean") -> Tensor:
    """LabelSmoothSoftmaxCEV1.forward (lit_models/utils.py:42-66): target = eps/C everywhere, overwritten with 1-eps at the label; rows whose
    label == ignore_index (:49-52) give 0 (:58); 'mean' = sum / n_valid (:59-60), 'sum' (:61-62), anything else per-row."""
    C = logits.shape[1]
    ignore = label == ignore_index
    n_valid = (~ignore).sum()
    lab = label.masked_fill(ignore, 0)
    tgt = torch.full_like(logits, eps / C)
    tgt.scatter_(1, lab[:, None], 1.0 - eps)
    rows = -(torch.log_softmax(logits, dim=1) * tgt).sum(1)
    rows = rows.masked_fill(ignore, 0.0)
    if reduction == "mean":
        return rows.sum() / n_valid
    if reduction == "sum":
        return rows.sum()
    return rows


def relaxation_loss(trans: Tensor, rel_idx: Tensor, q_head_idx: Tensor, a_head_idx: Tensor) -> Tensor:
    """lit_models/transformer.py:103-108."""
    ar = torch.arange(trans.shape[0])
    r0, r1 = trans[ar, rel_idx[:, 0]], trans[ar, rel_idx[:, 1]]
    qh, ah = trans[ar, q_head_idx], trans[ar, a_head_idx]
    return (F.relu(F.cosine_similarity(qh, ah)) + 1 - F.cosine_similarity(r0, r1)).mean(0)


def finetune_loss(sd, trans, input_ids, label, rel_idx, q_head_idx, a_head_idx, analogy_entity_ids,
                  mask_token_id: int = 103, alpha: float = 0.43, eps: float = 0.1):
    """Fine-tune branch of training_step (lit_models/transformer.py:92-109). Returns (loss, mask_logits)."""
    B = input_ids.shape[0]
    _, mask_idx = (input_ids == mask_token_id).nonzero(as_tuple=True)
    mask_logits = score(sd, trans[torch.arange(B), mask_idx], analogy_entity_ids)
    loss = label_smooth_ce(mask_logits, label, eps) + alpha * relaxation_loss(trans, rel_idx, q_head_idx, a_head_idx)
    return loss, mask_logits


def pretrain_loss(sd, trans, input_ids, label, pre_type, ent_range, rel_range, mask_token_id: int = 103, eps: float = 0.1):
    """Pre-train branch of training_step (lit_models/transformer.py:72-90)."""
    B = input_ids.shape[0]
    _, mask_idx = (input_ids == mask_token_id).nonzero(as_tuple=True)
    assert mask_idx.shape[0] == B, "only one mask in sequence!"
    rows = trans[torch.arange(B), mask_idx]
    loss = 0
    em = (pre_type != 2).nonzero(as_tuple=True)[0]
    if len(em) > 0:
        loss = loss + label_smooth_ce(score(sd, rows[em], slice(*ent_range)), label[em], eps)
    rm = (pre_type == 2).nonzero(as_tuple=True)[0]
    if len(rm) > 0:
        loss = loss + label_smooth_ce(score(sd, rows[rm], slice(*rel_range)), label[rm], eps)
    return loss


def ranks_double_sort(logits: Tensor, label: Tensor) -> np.ndarray:
    """lit_models/transformer.py:162-164: rank = argsort(argsort(-logits))[label] + 1."""
    _, o1 = torch.sort(logits, dim=1, descending=True)
    _, o2 = torch.sort(o1, dim=1)
    return (o2[torch.arange(logits.shape[0]), label].detach().cpu() + 1).numpy()


def ranks_count(logits: Tensor, label: Tensor) -> np.ndarray:
    """Tie-free equivalent of ``ranks_double_sort``: 1 + #(logit > logit[label])."""
    ll = logits[torch.arange(logits.shape[0]), label][:, None]
    return ((logits > ll).sum(1) + 1).cpu().numpy()


def rank_metrics(entity_ranks: np.ndarray) -> Dict[str, float]:
    """validation_epoch_end (lit_models/transformer.py:173-193)."""
    r = np.asarray(entity_ranks)
    m = {f"Eval_entity/hits{k}": float((r <= k).mean()) for k in (1, 3, 5, 10, 20)}
    m["Eval_entity/mean_rank"] = float(r.mean())
    m["Eval_entity/mrr"] = float((1.0 / r).mean())
    m["entity_hits10"] = m["Eval_entity/hits10"]
    m["entity_hits1"] = m["Eval_entity/hits1"]
    return m


# --------------------------------------------------------------------------- optimizer
NO_DECAY = ("bias", "LayerNorm.weight")


def decay_of(name: str, weight_decay: float = 0.01) -> float:
    """configure_optimizers grouping quirk (lit_models/transformer.py:225-230): substring match on
    'bias' / 'LayerNorm.weight' only, so CLIP layer_norm{1,2}.weight, vision_pre_layrnorm.weight,
    adaptive_weight.* and class_embedding ARE decayed."""
    return 0.0 if any(nd in name for nd in NO_DECAY) else weight_decay


def linear_schedule(step: int, warmup: float, total: int) -> float:
    """HF get_linear_schedule_with_warmup lambda (transformers==4.19.0, un-vendored; call site
    lit_models/transformer.py:233): warmup is the float 0.1*T."""
    if step < warmup:
        return float(step) / float(max(1, warmup))
    return max(0.0, float(total - step) / float(max(1, total - warmup)))


def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, wd: float,
               b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8) -> None:
    """torch.optim.AdamW update (torch==1.7.0 semantics, call site lit_models/transformer.py:232), in place."""
    p.mul_(1 - lr * wd)
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


# --------------------------------------------------------------------------- synthetic batch
def synthetic_batch(B: int, L: int, vc: VisionCfg, seed: int = 1234, n_entities: int = 11292,
                    n_analogy: int = 2063, base_vocab: int = 30522, n_rel: int = 192,
                    pretrain: bool = False) -> Dict[str, Tensor]:
    """MARS-shaped synthetic batch.  Layout
    [CLS] E_h d.. [SEP] [R] [SEP] E_t d.. [SEP] E_q d.. [SEP] [R] [SEP] [MASK] [SEP] [PAD]..
    This mirrors bench-side generation in mkg_analogy_amd.data_synth (kept separate on purpose:
    the product never imports the oracle)."""
    rng = np.random.default_rng(seed)
    R_TOK = base_vocab + n_entities + n_rel
    desc_lo = min(1000, base_vocab // 2)                               # description token ids live in [desc_lo, base_vocab)
    ids = np.zeros((B, L), np.int64)
    am = np.zeros((B, L), np.int64)
    tt = np.zeros((B, L), np.int64)
    sep = np.zeros((B, 6), np.int64)
    rel = np.zeros((B, 2), np.int64)
    qh = np.zeros(B, np.int64)
    ah = np.zeros(B, np.int64)
    for b in range(B):
        real = int(rng.integers(min(40, L - 4), L + 1))
        fixed = 13                                                    # specials + 3 entity tokens
        free = max(real - fixed, 0)
        cuts = np.sort(rng.integers(0, free + 1, size=2))
        dl = [cuts[0], cuts[1] - cuts[0], free - cuts[1]]
        ent = rng.integers(base_vocab, base_vocab + n_entities, size=3)
        toks: List[int] = [101]
        seps: List[int] = []
        rels: List[int] = []
        toks.append(int(ent[0])); q_pos = len(toks) - 1
        toks += list(rng.integers(desc_lo, base_vocab, size=dl[0]))
        toks.append(102); seps.append(len(toks) - 1)
        toks.append(R_TOK); rels.append(len(toks) - 1)
        toks.append(102); seps.append(len(toks) - 1)
        toks.append(int(ent[1]))
        toks += list(rng.integers(desc_lo, base_vocab, size=dl[1]))
        toks.append(102); seps.append(len(toks) - 1)
        toks.append(int(ent[2])); a_pos = len(toks) - 1
        toks += list(rng.integers(desc_lo, base_vocab, size=dl[2]))
        toks.append(102); seps.append(len(toks) - 1)
        toks.append(R_TOK); rels.append(len(toks) - 1)
        toks.append(102); seps.append(len(toks) - 1)
        toks.append(103)
        toks.append(102); seps.append(len(toks) - 1)
        n = len(toks)
        assert n <= L, (n, L)
        ids[b, :n] = toks
        am[b, :n] = 1
        tt[b, seps[2] + 1:n] = 1
        sep[b] = seps
        rel[b] = rels
        qh[b], ah[b] = q_pos, a_pos
    S = vc.image_size
    pix = rng.standard_normal((B, 2, 3, S, S), dtype=np.float32)
    drop = rng.random(B) < 0.4
    pix[drop, 1] = 0.0
    analogy = np.sort(rng.choice(np.arange(base_vocab, base_vocab + n_entities), size=n_analogy, replace=False))
    batch = dict(input_ids=ids, attention_mask=am, token_type_ids=tt, pixel_values=pix,
                 sep_idx=sep, rel_idx=rel, q_head_idx=qh, a_head_idx=ah,
                 label=rng.integers(0, n_analogy, size=B), rel_label=rng.integers(0, 27, size=B))
    out = {k: torch.from_numpy(np.asarray(v)) for k, v in batch.items()}
    out["analogy_entity_ids"] = torch.from_numpy(analogy)
    if pretrain:
        out.pop("sep_idx")
        out["pre_type"] = torch.from_numpy(rng.integers(1, 3, size=B))
        lab = np.where(out["pre_type"].numpy() == 2, rng.integers(0, n_rel, size=B), rng.integers(0, n_entities, size=B))
        out["label"] = torch.from_numpy(lab)
    return out


# --------------------------------------------------------------------------- vocabulary surgery
def init_relation_word(sd: Dict[str, Tensor], analogy_relation_ids: Sequence[int]) -> Dict[str, Tensor]:
    """TransformerLitModel._init_relation_word (lit_models/transformer.py:41-54) on a state dict:
    append the ``[R]`` row = mean of the analogy-relation rows (resize :822-836 copies old rows),
    zero-pad the decoder bias (modeling_unimo.py:915-924).  Returns a new dict."""
    out = dict(sd)
    W = sd["unimo.text_embeddings.word_embeddings.weight"]
    r = W[torch.as_tensor(list(analogy_relation_ids), dtype=torch.long)].mean(0, keepdim=True)
    out["unimo.text_embeddings.word_embeddings.weight"] = torch.cat([W, r], 0)
    b = sd["cls.predictions.bias"]
    out["cls.predictions.bias"] = torch.cat([b, b.new_zeros(1)], 0)
    return out
