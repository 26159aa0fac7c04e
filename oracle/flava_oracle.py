"""CPU oracle for the FLAVA backbone under the MarT framework (BASELINE config 4).

TEST INFRASTRUCTURE ONLY (same rules as mkgformer_oracle.py).  fp32 torch-CPU restatement of
``FlavaForMaskedLM.forward`` as MarT calls it (MarT/models/modeling_flava.py:2150-2204 -> FlavaModel.forward :1373-1476),
functional over the reference's parameter names.  Pinned by ``tests/golden/g5_flava_tiny.npz`` captured from the unmodified
reference import (oracle/gen_goldens.py:g5; extra shims listed there).  Third-party arithmetic: the additive mask helper
``get_extended_attention_mask`` lives in un-vendored transformers==4.19.0 ((1-mask)*-10000.0); it is restated here and the
golden run overrides the installed 5.x helper with exactly that formula -- parity at that boundary is pinned
to the version number only.

Everything the MarT path never reads is NOT evaluated: poolers, image/text final layernorms (only ``size(1)`` of the text
embeddings is used, :2187-2188), contrastive projections, logit_scale, codebook, ITM / MIM heads.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .mkgformer_oracle import _lin, _ln, _merge_heads, _split_heads, extended_mask, gelu_erf

Tensor = torch.Tensor


@dataclass
class FlavaCfg:
    """facebook/flava-full defaults (FlavaConfig()); all dropouts are 0.0 in that config."""
    vocab_size: int = 30522
    hidden_size: int = 768
    text_layers: int = 12
    image_layers: int = 12
    mm_layers: int = 6
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    max_position_embeddings: int = 512
    type_vocab_size: int = 2
    image_size: int = 224
    patch_size: int = 16
    layer_norm_eps: float = 1e-12

    @property
    def num_patches(self) -> int:
        return (self.image_size // self.patch_size) ** 2


def _layer_shapes(prefix: str, H: int, I: int) -> Dict[str, Tuple[int, ...]]:
    s = {}
    for n in ("query", "key", "value"):
        s[prefix + f"attention.attention.{n}.weight"] = (H, H)
        s[prefix + f"attention.attention.{n}.bias"] = (H,)
    s[prefix + "attention.attention.adaptive_weight.0"] = (1,)
    s[prefix + "attention.attention.adaptive_weight.1"] = (1,)
    s[prefix + "attention.output.dense.weight"] = (H, H)
    s[prefix + "attention.output.dense.bias"] = (H,)
    s[prefix + "intermediate.dense.weight"] = (I, H)
    s[prefix + "intermediate.dense.bias"] = (I,)
    s[prefix + "output.dense.weight"] = (H, I)
    s[prefix + "output.dense.bias"] = (H,)
    for ln in ("layernorm_before", "layernorm_after"):
        s[prefix + ln + ".weight"] = (H,)
        s[prefix + ln + ".bias"] = (H,)
    return s


def param_shapes(c: FlavaCfg) -> Dict[str, Tuple[int, ...]]:
    """Reference ``named_parameters()`` of FlavaForMaskedLM in construction order (modeling_flava.py:1227-1260: text_model,
    image_model, multimodal_model, projections; :2127-2133 head).  The decoder weight is tied to the word embedding."""
    H, I = c.hidden_size, c.intermediate_size
    s: Dict[str, Tuple[int, ...]] = {}
    s["flava.logit_scale"] = ()                    # FlavaModel's own parameter comes first in named_parameters()
    t = "flava.text_model."
    s[t + "embeddings.word_embeddings.weight"] = (c.vocab_size, H)
    s[t + "embeddings.position_embeddings.weight"] = (c.max_position_embeddings, H)
    s[t + "embeddings.token_type_embeddings.weight"] = (c.type_vocab_size, H)
    s[t + "embeddings.LayerNorm.weight"] = (H,)
    s[t + "embeddings.LayerNorm.bias"] = (H,)
    for i in range(c.text_layers):
        s.update(_layer_shapes(t + f"encoder.layer.{i}.", H, I))
    s[t + "layernorm.weight"] = (H,)
    s[t + "layernorm.bias"] = (H,)
    s[t + "pooler.dense.weight"] = (H, H)
    s[t + "pooler.dense.bias"] = (H,)
    m = "flava.image_model."
    s[m + "embeddings.cls_token"] = (1, 1, H)
    s[m + "embeddings.mask_token"] = (1, 1, H)
    s[m + "embeddings.position_embeddings"] = (1, c.num_patches + 1, H)
    s[m + "embeddings.patch_embeddings.projection.weight"] = (H, 3, c.patch_size, c.patch_size)
    s[m + "embeddings.patch_embeddings.projection.bias"] = (H,)
    for i in range(c.image_layers):
        s.update(_layer_shapes(m + f"encoder.layer.{i}.", H, I))
    s[m + "layernorm.weight"] = (H,)
    s[m + "layernorm.bias"] = (H,)
    s[m + "pooler.dense.weight"] = (H, H)
    s[m + "pooler.dense.bias"] = (H,)
    u = "flava.multimodal_model."
    s[u + "cls_token"] = (1, 1, H)
    for i in range(c.mm_layers):
        s.update(_layer_shapes(u + f"encoder.layer.{i}.", H, I))
    s[u + "layernorm.weight"] = (H,)
    s[u + "layernorm.bias"] = (H,)
    s[u + "pooler.dense.weight"] = (H, H)
    s[u + "pooler.dense.bias"] = (H,)
    for n in ("image_projection", "text_projection", "image_to_mm_projection", "text_to_mm_projection"):
        s[f"flava.{n}.weight"] = (H, H)
        s[f"flava.{n}.bias"] = (H,)
    s["cls.bias"] = (c.vocab_size,)
    s["cls.transform.dense.weight"] = (H, H)
    s["cls.transform.dense.bias"] = (H,)
    s["cls.transform.LayerNorm.weight"] = (H,)
    s["cls.transform.LayerNorm.bias"] = (H,)
    return s


def init_params(c: FlavaCfg, seed: int = 0) -> Dict[str, Tensor]:
    rng = np.random.default_rng(seed)
    out: Dict[str, Tensor] = {}
    for name, shape in param_shapes(c).items():
        if name.endswith("adaptive_weight.0"):
            v = np.full(shape, 0.25, np.float32)
        elif name.endswith("adaptive_weight.1"):
            v = np.full(shape, 0.5, np.float32)
        elif name == "flava.logit_scale":
            v = np.float32(2.6592)
        elif "LayerNorm.weight" in name or "layernorm" in name and name.endswith(".weight"):
            v = (1.0 + 0.05 * rng.standard_normal(shape)).astype(np.float32)
        else:
            v = (0.02 * rng.standard_normal(shape)).astype(np.float32)
        out[name] = torch.from_numpy(np.asarray(v, np.float32))
    return out


# ------------------------------------------------------------------------------------------- forward
def image_embed(sd, c: FlavaCfg, pixel_values: Tensor) -> Tensor:
    """FlavaImageEmbeddings.forward (modeling_flava.py:308-343).  Quirk reproduced: the second image gets
    ``position_embeddings[:, :P]`` i.e. INCLUDING the cls position (:338)."""
    p = "flava.image_model.embeddings."
    w, b = sd[p + "patch_embeddings.projection.weight"], sd[p + "patch_embeddings.projection.bias"]
    B = pixel_values.shape[0]
    head = F.conv2d(pixel_values[:, 0], w, b, stride=c.patch_size).flatten(2).transpose(1, 2)
    tail = F.conv2d(pixel_values[:, 1], w, b, stride=c.patch_size).flatten(2).transpose(1, 2)
    pos = sd[p + "position_embeddings"]
    x = torch.cat([sd[p + "cls_token"].expand(B, -1, -1), head], dim=1) + pos
    tail = tail + pos[:, :tail.shape[1]]
    return torch.cat([x, tail], dim=1)


def text_embed(sd, c: FlavaCfg, input_ids: Tensor, token_type_ids: Tensor) -> Tensor:
    """FlavaTextEmbeddings.forward (modeling_flava.py:406-438)."""
    p = "flava.text_model.embeddings."
    L = input_ids.shape[1]
    x = sd[p + "word_embeddings.weight"][input_ids] + sd[p + "token_type_embeddings.weight"][token_type_ids]
    x = x + sd[p + "position_embeddings.weight"][:L][None]
    return _ln(x, sd[p + "LayerNorm.weight"], sd[p + "LayerNorm.bias"], c.layer_norm_eps)


def flava_reweight_factor(w0: Tensor, w1: Tensor, sep: Tensor, L: int) -> Tensor:
    """modeling_flava.py:494-496: rows 1..s-1 x cols >= s scaled by clamp(w0,0,.5) (row 0 = [CLS] is NOT touched),
    rows >= s x cols >= s by clamp(w1,.5,1)."""
    c0, c1 = torch.clamp(w0, 0, 0.5), torch.clamp(w1, 0.5, 1)
    ar = torch.arange(L)
    s = sep[:, None]
    col_hi = (ar[None, :] >= s)[:, None, :]
    row_mid = ((ar[None, :] >= 1) & (ar[None, :] < s))[:, :, None]
    row_hi = (ar[None, :] >= s)[:, :, None]
    one = torch.ones(())
    f = torch.where(col_hi & row_mid, c0, torch.where(col_hi & row_hi, c1, one))
    return f[:, None]


def layer(sd, c: FlavaCfg, prefix: str, x: Tensor, ext_mask: Optional[Tensor], sep_idx: Optional[Tensor]) -> Tensor:
    """FlavaLayer.forward (modeling_flava.py:635-665) with FlavaSelfAttention (:472-521); dropouts are 0 in the config."""
    nh = c.num_attention_heads
    dh = c.hidden_size // nh
    L = x.shape[1]
    h = _ln(x, sd[prefix + "layernorm_before.weight"], sd[prefix + "layernorm_before.bias"], c.layer_norm_eps)
    q = _split_heads(_lin(h, sd, prefix + "attention.attention.query"), nh)
    k = _split_heads(_lin(h, sd, prefix + "attention.attention.key"), nh)
    v = _split_heads(_lin(h, sd, prefix + "attention.attention.value"), nh)
    scores = (q @ k.transpose(-1, -2)) / (dh ** 0.5)
    if sep_idx is not None:
        scores = scores * flava_reweight_factor(sd[prefix + "attention.attention.adaptive_weight.0"],
                                                sd[prefix + "attention.attention.adaptive_weight.1"], sep_idx[:, 2], L)
    if ext_mask is not None:
        scores = scores + ext_mask
    ctx = _merge_heads(torch.softmax(scores, dim=-1) @ v)
    x = _lin(ctx, sd, prefix + "attention.output.dense") + x
    h = _ln(x, sd[prefix + "layernorm_after.weight"], sd[prefix + "layernorm_after.bias"], c.layer_norm_eps)
    h = gelu_erf(_lin(h, sd, prefix + "intermediate.dense"))
    return _lin(h, sd, prefix + "output.dense") + x


def forward(sd, c: FlavaCfg, input_ids, attention_mask, token_type_ids, pixel_values, sep_idx=None, taps=None):
    """-> trans_hidden_states [B,L,H] (FlavaForMaskedLM.forward :2150-2204; scoring via mkgformer_oracle-style ``score``)."""
    img = image_embed(sd, c, pixel_values)
    for i in range(c.image_layers):
        img = layer(sd, c, f"flava.image_model.encoder.layer.{i}.", img, None, None)
        if taps is not None:
            taps[f"i{i}"] = img                      # per-layer outputs (golden G9b holds the reference's for layers 0 / 11)
    txt = text_embed(sd, c, input_ids, token_type_ids)
    em = extended_mask(attention_mask)
    for i in range(c.text_layers):
        txt = layer(sd, c, f"flava.text_model.encoder.layer.{i}.", txt, em, sep_idx)
        if taps is not None:
            taps[f"t{i}"] = txt
    if taps is not None:
        taps["img"], taps["txt"] = img, txt
    # projections of the PRE-final-layernorm last hidden states (:1430,1450)
    mm = torch.cat([_lin(img, sd, "flava.image_to_mm_projection"), _lin(txt, sd, "flava.text_to_mm_projection")], dim=1)
    mm = torch.cat([sd["flava.multimodal_model.cls_token"].expand(mm.shape[0], -1, -1), mm], dim=1)
    for i in range(c.mm_layers):
        mm = layer(sd, c, f"flava.multimodal_model.encoder.layer.{i}.", mm, None, None)    # all-ones mask -> additive zeros
        if taps is not None:
            taps[f"m{i}"] = mm
    mm = _ln(mm, sd["flava.multimodal_model.layernorm.weight"], sd["flava.multimodal_model.layernorm.bias"], c.layer_norm_eps)
    seq = mm[:, -input_ids.shape[1]:, :]
    t = gelu_erf(_lin(seq, sd, "cls.transform.dense"))
    return _ln(t, sd["cls.transform.LayerNorm.weight"], sd["cls.transform.LayerNorm.bias"], c.layer_norm_eps)


def score(sd, trans_rows: Tensor, ids) -> Tensor:
    W, b = sd["flava.text_model.embeddings.word_embeddings.weight"], sd["cls.bias"]
    ids = torch.as_tensor(ids, dtype=torch.long)
    return F.linear(trans_rows, W[ids], b[ids])
