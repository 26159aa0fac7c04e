"""Manual forward/backward schedule of the FLAVA path MarT runs (reference: MarT/models/modeling_flava.py:1373-1476
FlavaModel.forward, :2150-2204 FlavaForMaskedLM.forward) on the HIP kernels -- SURVEY 8(a) row 19.

Three stacks of the same pre-LN block (FlavaLayer.forward :635-665: x + Attn(LN x); x + MLP(LN x), erf GELU):
  image  12 layers over 1+P+P tokens (no mask),
  text   12 layers over L tokens (additive padding mask, FLAVA variant of the adaptive analogy reweight: rows 1..s-1),
  multimodal 6 layers over 1 + (1+2P) + L tokens (all-ones mask),
joined by the two *_to_mm projections of the PRE-final-layernorm stream outputs.  Dropouts are 0 in this configuration.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import os

import torch

from . import ops
from .params import FlatStore

BF, F32, HF = torch.bfloat16, torch.float32, torch.float16
WORD = "flava.text_model.embeddings.word_embeddings.weight"


def _e(shape, dtype, dev):
    return torch.empty(shape, device=dev, dtype=dtype)


def _layer_names(p: str) -> List[str]:
    a = p + "attention.attention."
    return ([a + f"{n}.weight" for n in ("query", "key", "value")] + [a + f"{n}.bias" for n in ("query", "key", "value")] +
            [a + "adaptive_weight.0", a + "adaptive_weight.1", p + "attention.output.dense.weight", p + "attention.output.dense.bias",
             p + "layernorm_before.weight", p + "layernorm_before.bias", p + "intermediate.dense.weight", p + "intermediate.dense.bias",
             p + "output.dense.weight", p + "output.dense.bias", p + "layernorm_after.weight", p + "layernorm_after.bias"])


def flava_layout_order(nt: int, ni: int, nm: int) -> List[str]:
    """Backward-completion order: head, multimodal stack (top down), projections, text stack, image stack, embeddings,
    never-used tensors, tied word embedding last."""
    o = ["cls.transform.dense.weight", "cls.transform.dense.bias", "cls.transform.LayerNorm.weight", "cls.transform.LayerNorm.bias",
         "flava.multimodal_model.layernorm.weight", "flava.multimodal_model.layernorm.bias"]
    for l in reversed(range(nm)):
        o += _layer_names(f"flava.multimodal_model.encoder.layer.{l}.")
    o += ["flava.multimodal_model.cls_token", "flava.image_to_mm_projection.weight", "flava.image_to_mm_projection.bias",
          "flava.text_to_mm_projection.weight", "flava.text_to_mm_projection.bias"]
    for l in reversed(range(nt)):
        o += _layer_names(f"flava.text_model.encoder.layer.{l}.")
    for l in reversed(range(ni)):
        o += _layer_names(f"flava.image_model.encoder.layer.{l}.")
    t, m = "flava.text_model.embeddings.", "flava.image_model.embeddings."
    o += [t + "LayerNorm.weight", t + "LayerNorm.bias", t + "position_embeddings.weight", t + "token_type_embeddings.weight",
          m + "cls_token", m + "position_embeddings", m + "patch_embeddings.projection.weight", m + "patch_embeddings.projection.bias",
          m + "mask_token", "flava.logit_scale"]
    for mod in ("text_model", "image_model"):
        o += [f"flava.{mod}.layernorm.weight", f"flava.{mod}.layernorm.bias"]
    for mod in ("text_model", "image_model", "multimodal_model"):
        o += [f"flava.{mod}.pooler.dense.weight", f"flava.{mod}.pooler.dense.bias"]
    o += ["flava.image_projection.weight", "flava.image_projection.bias", "flava.text_projection.weight", "flava.text_projection.bias",
          "cls.bias", WORD]
    return o


def flava_f16_weight(name: str) -> bool:
    """GEMM weights of the three encoder stacks: the tensors that get an fp16 forward shadow (FlavaEngine.f16 / .f16_img, as engine.UnimoEngine.text_f16)."""
    if name in ("flava.image_model.embeddings.patch_embeddings.projection.weight", "flava.image_to_mm_projection.weight", "flava.text_to_mm_projection.weight"):
        return True
    return (name.startswith(("flava.text_model.encoder.layer.", "flava.multimodal_model.encoder.layer.", "flava.image_model.encoder.layer."))
            and name.endswith(".weight") and "layernorm" not in name)


def FLAVA_DEAD(nl: Tuple[int, int, int]) -> Tuple[str, ...]:
    """Tensors whose gradient is None in the reference's MarT usage (golden g5: 26 names at 3/3/2 layers)."""
    nt, ni, nm = nl
    d = ["flava.logit_scale", "flava.image_model.embeddings.mask_token", "flava.text_model.layernorm.", "flava.image_model.layernorm.",
         "flava.text_model.pooler.", "flava.image_model.pooler.", "flava.multimodal_model.pooler.", "flava.image_projection.",
         "flava.text_projection."]
    d += [f"flava.image_model.encoder.layer.{l}.attention.attention.adaptive_weight." for l in range(ni)]
    d += [f"flava.multimodal_model.encoder.layer.{l}.attention.attention.adaptive_weight." for l in range(nm)]
    return tuple(d)


def flava_gemm_groups(nt: int, ni: int, nm: int):
    g = [("head", ("cls.transform.dense.weight",)), ("i2m", ("flava.image_to_mm_projection.weight",)),
         ("t2m", ("flava.text_to_mm_projection.weight",))]
    for tag, mod, n in (("t", "text_model", nt), ("i", "image_model", ni), ("m", "multimodal_model", nm)):
        for l in range(n):
            p = f"flava.{mod}.encoder.layer.{l}."
            g += [(f"{tag}{l}.qkv", tuple(p + f"attention.attention.{x}.weight" for x in ("query", "key", "value"))),
                  (f"{tag}{l}.o", (p + "attention.output.dense.weight",)), (f"{tag}{l}.fc1", (p + "intermediate.dense.weight",)),
                  (f"{tag}{l}.fc2", (p + "output.dense.weight",))]
    return g


class FlavaEngine:
    def __init__(self, store: FlatStore, config):
        self.st, self.cfg = store, config
        tc, ic, mc = config.text_config, config.image_config, config.multimodal_config
        self.H, self.nh, self.I = tc.hidden_size, tc.num_attention_heads, tc.intermediate_size
        assert ic.hidden_size == self.H and mc.hidden_size == self.H and self.H // self.nh == 64, "HIP path: equal widths, head_dim 64"
        self.nt, self.ni, self.nm = tc.num_hidden_layers, ic.num_hidden_layers, mc.num_hidden_layers
        self.eps = float(tc.layer_norm_eps)
        self.grad_ready: Optional[Callable[[int], None]] = None
        self.taps: Optional[dict] = None
        self.head_split = os.environ.get("MART_HEAD_SPLIT", "1") == "1"   # head transform + scoring GEMM on two-term operand splits (engine.UnimoEngine.head_split)
        # the forward linear layers of the TEXT and MULTIMODAL stacks on fp16 operands (same switch and reasoning as engine.UnimoEngine.text_f16:
        # the logits are read off the multimodal stack's text positions; the image stack -- 2/3 of the FLOPs -- stays bf16)
        self.f16 = os.environ.get("MART_TEXT_F16", "1") == "1" and getattr(store, "f16_weight", None) is flava_f16_weight
        # round 5: the IMAGE stack's forward products on fp16 operands as well (every operand is a LayerNorm output, an attention context or a GELU
        # output; same MFMA rate, 8 x finer rounding): VERDICT r4 item 7 -- the image stack was the largest remaining term of the bf16 logit error
        # (G9 / G9b at 7.7e-3 / 8.3e-3 of a 1e-2 budget).  Costs the fp16 twins' writes in a training step (+1.1 GB per layer at B = 256)
        self.f16_img = self.f16 and os.environ.get("MART_IMAGE_F16", "1") == "1"
        self.grad_stream_bf16 = os.environ.get("MART_GRAD_STREAM_BF16", "1") == "1"    # the residual GRADIENT stream in bf16 inside a stack (_layer_bwd)

    # ------------------------------------------------------------------ one pre-LN block
    def _layer_fwd(self, p: str, x, M: int, attn_kw: dict, want_bf16: bool, f16: bool = False):
        """``f16``: the four products multiply fp16 operands (LayerNorm outputs, attention context and GELU output written as fp16 twins, fp16
        weight shadow); the bf16 twins are kept only when a backward pass will read them."""
        st, H, I, dev = self.st, self.H, self.I, x.device
        keep = bool(getattr(self, "save_for_backward", True))      # False under torch.no_grad(): backward-only outputs are skipped
        kb = keep or not f16                                        # bf16 twins needed
        W = (lambda *n: st.h(*n)) if f16 else (lambda *n: st.fused(list(n)))
        h1, m1, r1 = (_e((M, H), BF, dev) if kb else None), _e((M,), F32, dev), _e((M,), F32, dev)
        h1h = _e((M, H), HF, dev) if f16 else None
        ops.ln_fwd(x_f32=x, gamma=st.m(p + "layernorm_before.weight"), beta=st.m(p + "layernorm_before.bias"), eps=self.eps, M=M, H=H,
                   mean=m1, rstd=r1, out_bf16=h1, out_f16=h1h)
        a = p + "attention.attention."
        qkv = _e((M, 3 * H), BF, dev)
        ops.gemm_nt(h1h if f16 else h1, W(*[a + f"{n}.weight" for n in ("query", "key", "value")]), qkv,
                    bias=st.fused([a + f"{n}.bias" for n in ("query", "key", "value")], st.master))
        ctx = _e((M, H), BF, dev)
        ctxh = _e((M, H), HF, dev) if f16 else None
        lse = _e((attn_kw["B"], self.nh, attn_kw["Sq"]), F32, dev)
        kw = dict(q=qkv[:, :H], k=qkv[:, H:2 * H], v=qkv[:, 2 * H:], ctx=ctx, lse=lse, nh=self.nh, scale=0.125, **attn_kw)
        ops.attn_fwd(ctx_f16=ctxh, **kw)
        x1 = _e((M, H), F32, dev)
        ops.gemm_nt(ctxh if f16 else ctx, W(p + "attention.output.dense.weight"), x1, bias=st.m(p + "attention.output.dense.bias"), res_f32=x)
        h2, m2, r2 = (_e((M, H), BF, dev) if kb else None), _e((M,), F32, dev), _e((M,), F32, dev)
        h2h = _e((M, H), HF, dev) if f16 else None
        ops.ln_fwd(x_f32=x1, gamma=st.m(p + "layernorm_after.weight"), beta=st.m(p + "layernorm_after.bias"), eps=self.eps, M=M, H=H,
                   mean=m2, rstd=r2, out_bf16=h2, out_f16=h2h)
        z, f = (_e((M, I), BF, dev) if keep else None), (_e((M, I), BF, dev) if kb else None)   # z = gelu'(intermediate.dense output)
        fh = _e((M, I), HF, dev) if f16 else None
        ops.gemm_nt(h2h if f16 else h2, W(p + "intermediate.dense.weight"), fh if f16 else f, bias=st.m(p + "intermediate.dense.bias"), act=ops.ACT_GELU,
                    preact=z, preact_grad=keep, C2=f if f16 else None)
        x2 = _e((M, H), F32, dev)
        x2b = _e((M, H), BF, dev) if want_bf16 else None
        ops.gemm_nt(fh if f16 else f, W(p + "output.dense.weight"), x2, bias=st.m(p + "output.dense.bias"), res_f32=x1, C2=x2b)
        return x2, x2b, dict(x=x, m1=m1, r1=r1, h1=h1, qkv=qkv, kw=kw, x1=x1, m2=m2, r2=r2, h2=h2, z=z, f=f)

    def _layer_bwd(self, p: str, key: str, s: dict, M: int, dx, dxb, dw=None, want_f32: bool = True):
        """dx (f32) / dxb (bf16) hold d(loss)/d(layer output) on entry and d(loss)/d(layer input) on return (in place).
        grad_stream_bf16 (round 6, as engine.UnimoEngine): inside a stack the gradient w.r.t. the residual stream travels as dxb alone -- the LayerNorm
        backward reads the bf16 residual operand and writes the bf16 total (10 bytes per element instead of 16); dx is read by nothing and rewritten only
        where the caller needs the f32 tensor again (``want_f32``: the stack's first layer, whose input gradient feeds the embedding / projection code)."""
        st, H, I, dev = self.st, self.H, self.I, dxb.device
        gb = self.grad_stream_bf16

        def wgrad(X, Y, wn, bn):
            g = st.g(wn)
            ops.gemm_tn(X, Y, g.view(g.shape[0], -1), colsum=st.g(bn))
        wgrad(dxb, s["f"], p + "output.dense.weight", p + "output.dense.bias")
        dz = _e((M, I), BF, dev)
        ops.gemm_nt(dxb, st.wt(key + ".fc2"), dz, mulz=s["z"], mul_act=ops.ACT_STORED)
        wgrad(dz, s["h2"], p + "intermediate.dense.weight", p + "intermediate.dense.bias")
        dh2 = _e((M, H), BF, dev)
        ops.gemm_nt(dz, st.wt(key + ".fc1"), dh2)
        del dz
        dx1, dx1b = (None if gb else _e((M, H), F32, dev)), _e((M, H), BF, dev)
        ops.ln_bwd(dy_bf16=dh2, s=s["x1"], mean=s["m2"], rstd=s["r2"], gamma=st.m(p + "layernorm_after.weight"), M=M, H=H, add_f32=None if gb else dx,
                   add_bf16=dxb if gb else None, ds_f32=dx1, ds_bf16=dx1b, bf16_total=True, dgamma=st.g(p + "layernorm_after.weight"), dbeta=st.g(p + "layernorm_after.bias"))
        kw = s["kw"]
        wgrad(dx1b, kw["ctx"], p + "attention.output.dense.weight", p + "attention.output.dense.bias")
        dctx = dh2
        ops.gemm_nt(dx1b, st.wt(key + ".o"), dctx)
        dqkv = _e((M, 3 * H), BF, dev)
        delta = _e((kw["B"], self.nh, kw["Sq"]), F32, dev)
        ops.attn_bwd(dctx=dctx, delta=delta, dq=dqkv[:, :H], dk=dqkv[:, H:2 * H], dv=dqkv[:, 2 * H:], dw=dw, **kw)
        a = p + "attention.attention."
        ops.gemm_tn(dqkv, s["h1"], st.fused([a + f"{n}.weight" for n in ("query", "key", "value")], st.grad),
                    colsum=st.fused([a + f"{n}.bias" for n in ("query", "key", "value")], st.grad))
        dh1 = dctx
        ops.gemm_nt(dqkv, st.wt(key + ".qkv"), dh1)
        ops.ln_bwd(dy_bf16=dh1, s=s["x"], mean=s["m1"], rstd=s["r1"], gamma=st.m(p + "layernorm_before.weight"), M=M, H=H, add_f32=dx1,
                   add_bf16=dx1b if gb else None, ds_f32=dx if (want_f32 or not gb) else None, ds_bf16=dxb, bf16_total=True, dgamma=st.g(p + "layernorm_before.weight"), dbeta=st.g(p + "layernorm_before.bias"))

    # ------------------------------------------------------------------ forward
    def forward(self, input_ids, attention_mask, token_type_ids, pixel_values, sep_idx, train: bool, seed: int,
                image_table=None, image_index=None):
        st, H, dev = self.st, self.H, input_ids.device
        B, Lq = input_ids.shape
        ic = self.cfg.image_config
        S, p = ic.image_size, ic.patch_size
        P = (S // p) ** 2
        Nv = 1 + 2 * P
        Sm = 1 + Nv + Lq
        Mi, Mt, Mm = B * Nv, B * Lq, B * Sm
        sv: Dict[str, object] = dict(B=B, L=Lq, P=P, Nv=Nv, Sm=Sm, ids=input_ids, tt=token_type_ids)
        # ---- image embeddings (FlavaImageEmbeddings.forward :308-343): conv(+bias) as GEMM, cls, positions (tail image: pos[:P])
        Kp = 3 * p * p
        e = "flava.image_model.embeddings."
        xi = _e((Mi, H), F32, dev)
        if self.f16:
            # patch embedding on fp16 operands with an f32 result: the bf16 rounding of the pixel patches and of the embedding itself (2^-9 each) was
            # the floor of every image tap (G9b: 2.9e-3 after layer 0, the text taps 2-5e-4).  The bf16 copy is the weight gradient's operand.
            assert st.f16_weight(e + "patch_embeddings.projection.weight")
            pf = _e((B * 2 * P, Kp), F32, dev)
            if image_index is not None:
                ops.patchify_f32(image_table, image_index.contiguous(), pf, B, S, p)
            else:
                pix = pixel_values.contiguous()
                assert pix.dtype == F32 and tuple(pix.shape) == (B, 2, 3, S, S)
                ops.patchify_f32(pix, None, pf, B, S, p)
            ph = _e((B * 2 * P, Kp), HF, dev)
            ops.cast_f32_f16(pf, ph)
            patches = None
            if bool(getattr(self, "save_for_backward", True)):
                patches = _e((B * 2 * P, Kp), BF, dev)
                ops.cast_f32_bf16(pf, patches)
            pe32 = _e((B * 2 * P, H), F32, dev)
            ops.gemm_nt(ph, st.h(e + "patch_embeddings.projection.weight").view(H, Kp), pe32, bias=st.m(e + "patch_embeddings.projection.bias"))
            ops.vision_assemble_f32(pe32, st.m(e + "cls_token"), st.m(e + "position_embeddings"), xi, B, P, H, tail_shift=1)
        else:
            patches = _e((B * 2 * P, Kp), BF, dev)
            if image_index is not None:
                ops.patchify_gather(image_table, image_index.contiguous(), patches, B, S, p)
            else:
                pix = pixel_values.contiguous()
                assert pix.dtype == F32 and tuple(pix.shape) == (B, 2, 3, S, S)
                ops.patchify(pix, patches, B, S, p)
            pe = _e((B * 2 * P, H), BF, dev)
            ops.gemm_nt(patches, st.w(e + "patch_embeddings.projection.weight").view(H, Kp), pe, bias=st.m(e + "patch_embeddings.projection.bias"))
            ops.vision_assemble(pe, st.m(e + "cls_token"), st.m(e + "position_embeddings"), xi, B, P, H, tail_shift=1)
        sv["patches"] = patches
        ikw = dict(B=B, Sq=Nv, Sk=Nv)
        for l in range(self.ni):
            xi, xib, sv[f"i{l}"] = self._layer_fwd(f"flava.image_model.encoder.layer.{l}.", xi, Mi, ikw, l == self.ni - 1, self.f16_img)
            if self.taps is not None:
                self.taps[f"i{l}"] = xi.view(B, Nv, H).clone()             # per-layer outputs (tests: golden G9b)
        # ---- text embeddings + stack (FlavaTextEmbeddings :406-438; FLAVA reweight :494-496)
        t = "flava.text_model.embeddings."
        s_t, tmean, trstd = _e((Mt, H), F32, dev), _e((Mt,), F32, dev), _e((Mt,), F32, dev)
        xt = _e((Mt, H), F32, dev)
        ops.text_embed_fwd(ids=input_ids, tt=token_type_ids, word=st.m(t + "word_embeddings.weight"), pos=st.m(t + "position_embeddings.weight"),
                           type_=st.m(t + "token_type_embeddings.weight"), gamma=st.m(t + "LayerNorm.weight"), beta=st.m(t + "LayerNorm.bias"),
                           eps=self.eps, p_drop=0.0, seed=0, B=B, Lq=Lq, H=H, s_out=s_t, mean=tmean, rstd=trstd, out_f32=xt, out_bf16=None)
        sv["temb"] = (s_t, tmean, trstd)
        for l in range(self.nt):
            pfx = f"flava.text_model.encoder.layer.{l}."
            a = pfx + "attention.attention."
            tkw = dict(B=B, Sq=Lq, Sk=Lq, attn_mask=attention_mask, sep=sep_idx[:, 2:] if sep_idx is not None else None,
                       sep_stride=sep_idx.shape[1] if sep_idx is not None else 0,
                       w0=st.m(a + "adaptive_weight.0") if sep_idx is not None else None,
                       w1=st.m(a + "adaptive_weight.1") if sep_idx is not None else None, rw_skip_row0=True)
            xt, xtb, sv[f"t{l}"] = self._layer_fwd(pfx, xt, Mt, tkw, l == self.nt - 1, self.f16)
            if self.taps is not None:
                self.taps[f"t{l}"] = xt.view(B, Lq, H).clone()
        if self.taps is not None:
            self.taps["img"], self.taps["txt"] = xi.view(B, Nv, H).clone(), xt.view(B, Lq, H).clone()
        # ---- multimodal input: [cls | image_to_mm(img) | text_to_mm(txt)]  (:1430,1450,1455-1456; cls :1182-1184)
        xm = _e((B, Sm, H), F32, dev)
        xm[:, 0, :].copy_(st.m("flava.multimodal_model.cls_token").view(1, H))
        if self.f16:
            # the projections read the UN-normalised last hidden states: as bf16 copies their 2^-9 rounding was the whole error of the first
            # multimodal tap (G9b m0: 2.4e-3 with every stack below it at 4-7e-4); fp16 copies + fp16 weight shadows, f32 result as before
            xih, xth = _e((Mi, H), HF, dev), _e((Mt, H), HF, dev)
            ops.cast_f32_f16(xi, xih)
            ops.cast_f32_f16(xt, xth)
            ops.gemm_nt(xih, st.h("flava.image_to_mm_projection.weight"), xm[0, 1:], bias=st.m("flava.image_to_mm_projection.bias"),
                        M=Nv, batch=B, stride_a=Nv * H, stride_c=Sm * H)
            ops.gemm_nt(xth, st.h("flava.text_to_mm_projection.weight"), xm[0, 1 + Nv:], bias=st.m("flava.text_to_mm_projection.bias"),
                        M=Lq, batch=B, stride_a=Lq * H, stride_c=Sm * H)
        else:
            ops.gemm_nt(xib, st.w("flava.image_to_mm_projection.weight"), xm[0, 1:], bias=st.m("flava.image_to_mm_projection.bias"),
                        M=Nv, batch=B, stride_a=Nv * H, stride_c=Sm * H)
            ops.gemm_nt(xtb, st.w("flava.text_to_mm_projection.weight"), xm[0, 1 + Nv:], bias=st.m("flava.text_to_mm_projection.bias"),
                        M=Lq, batch=B, stride_a=Lq * H, stride_c=Sm * H)
        sv["xib"], sv["xtb"] = xib, xtb
        xm = xm.view(Mm, H)
        mkw = dict(B=B, Sq=Sm, Sk=Sm)
        for l in range(self.nm):
            xm, _, sv[f"m{l}"] = self._layer_fwd(f"flava.multimodal_model.encoder.layer.{l}.", xm, Mm, mkw, False, self.f16)
            if self.taps is not None:
                self.taps[f"m{l}"] = xm.view(B, Sm, H).clone()
        # ---- final multimodal layernorm, text positions, MLM head transform (:1209, :2187-2188, :1676-1680)
        mm_b, mmean, mrstd = _e((Mm, H), BF, dev), _e((Mm,), F32, dev), _e((Mm,), F32, dev)
        split = self.head_split
        mm_f = _e((Mm, H), F32, dev) if split else None
        ops.ln_fwd(x_f32=xm, gamma=st.m("flava.multimodal_model.layernorm.weight"), beta=st.m("flava.multimodal_model.layernorm.bias"),
                   eps=self.eps, M=Mm, H=H, mean=mmean, rstd=mrstd, out_bf16=mm_b, out_f32=mm_f)
        rows = (torch.arange(B, device=dev, dtype=torch.int32)[:, None] * Sm + (1 + Nv) + torch.arange(Lq, device=dev, dtype=torch.int32)[None]).reshape(-1).contiguous()
        y, zh = _e((Mt, H), F32, dev), _e((Mt, H), BF, dev)
        if split:
            ver, w3 = getattr(self, "_head_w3", (-1, None))
            if ver != st.version:
                w3 = ops.split_bf16x3(st.m("cls.transform.dense.weight"), 1)
                self._head_w3 = (st.version, w3)
            ops.gemm_nt(ops.split_bf16x3_rows(mm_f, rows, 0), w3, y, bias=st.m("cls.transform.dense.bias"), act=ops.ACT_GELU, preact=zh)
        else:
            ops.gemm_nt(mm_b, st.w("cls.transform.dense.weight"), y, a_rows=rows, bias=st.m("cls.transform.dense.bias"), act=ops.ACT_GELU, preact=zh)
        trans, transb = _e((Mt, H), F32, dev), _e((Mt, H), BF, dev)
        hm, hr = _e((Mt,), F32, dev), _e((Mt,), F32, dev)
        ops.ln_fwd(x_f32=y, gamma=st.m("cls.transform.LayerNorm.weight"), beta=st.m("cls.transform.LayerNorm.bias"), eps=self.eps, M=Mt, H=H,
                   mean=hm, rstd=hr, out_f32=trans, out_bf16=transb)
        sv["head"] = (xm, mmean, mrstd, mm_b, rows, y, zh, hm, hr)
        return trans.view(B, Lq, H), transb, sv

    # ------------------------------------------------------------------ backward
    def backward(self, sv, dtrans: torch.Tensor) -> None:
        st, H, dev = self.st, self.H, dtrans.device
        B, Lq, P, Nv, Sm = sv["B"], sv["L"], sv["P"], sv["Nv"], sv["Sm"]
        Mi, Mt, Mm = B * Nv, B * Lq, B * Sm
        def notify(off):
            if self.grad_ready is not None:
                self.grad_ready(off)
            if getattr(self, "grad_ready_async", None) is not None:    # single-stream engine: nothing to wait for besides the current stream
                self.grad_ready_async(off, [])
        xm, mmean, mrstd, mm_b, rows, y, zh, hm, hr = sv["head"]
        # head transform
        dyb = _e((Mt, H), BF, dev)
        ops.ln_bwd(dy_f32=dtrans.contiguous().view(Mt, H), s=y, mean=hm, rstd=hr, gamma=st.m("cls.transform.LayerNorm.weight"), M=Mt, H=H, ds_bf16=dyb,
                   dgamma=st.g("cls.transform.LayerNorm.weight"), dbeta=st.g("cls.transform.LayerNorm.bias"))
        dzh = _e((Mt, H), BF, dev)
        ops.act_bwd(dyb, zh, ops.ACT_GELU, dzh)
        seq_b = _e((Mt, H), BF, dev)
        ops.gather_rows_bf16(mm_b, rows, seq_b)
        ops.gemm_tn(dzh, seq_b, st.g("cls.transform.dense.weight"), colsum=st.g("cls.transform.dense.bias"))
        dseq = _e((Mt, H), F32, dev)
        ops.gemm_nt(dzh, st.wt("head"), dseq)
        dmm_ln = torch.zeros((Mm, H), device=dev, dtype=F32)
        ops.scatter_add_rows_f32(dseq, rows, dmm_ln)
        # final multimodal layernorm
        dx, dxb = _e((Mm, H), F32, dev), _e((Mm, H), BF, dev)
        ops.ln_bwd(dy_f32=dmm_ln, s=xm, mean=mmean, rstd=mrstd, gamma=st.m("flava.multimodal_model.layernorm.weight"), M=Mm, H=H,
                   ds_f32=dx, ds_bf16=dxb, bf16_total=True, dgamma=st.g("flava.multimodal_model.layernorm.weight"),
                   dbeta=st.g("flava.multimodal_model.layernorm.bias"))
        for l in reversed(range(self.nm)):
            self._layer_bwd(f"flava.multimodal_model.encoder.layer.{l}.", f"m{l}", sv[f"m{l}"], Mm, dx, dxb, want_f32=(l == 0))
            sv[f"m{l}"] = None
        # split d(multimodal input): cls | image projection | text projection
        dx3, dxb3 = dx.view(B, Sm, H), dxb.view(B, Sm, H)
        st.g("flava.multimodal_model.cls_token").view(H).add_(dx3[:, 0, :].sum(0))
        xib, xtb = sv["xib"], sv["xtb"]
        ops.gemm_tn(dxb3[0, 1:], xib, st.g("flava.image_to_mm_projection.weight"), M=Nv, colsum=st.g("flava.image_to_mm_projection.bias"),
                    batch=B, stride_x=Sm * H, stride_y=Nv * H, stride_o=0, splits=1)
        ops.gemm_tn(dxb3[0, 1 + Nv:], xtb, st.g("flava.text_to_mm_projection.weight"), M=Lq, colsum=st.g("flava.text_to_mm_projection.bias"),
                    batch=B, stride_x=Sm * H, stride_y=Lq * H, stride_o=0, splits=1)
        dxi, dxib = _e((Mi, H), F32, dev), _e((Mi, H), BF, dev)
        ops.gemm_nt(dxb3[0, 1:], st.wt("i2m"), dxi, M=Nv, batch=B, stride_a=Sm * H, stride_c=Nv * H, C2=dxib)
        dxt, dxtb = _e((Mt, H), F32, dev), _e((Mt, H), BF, dev)
        ops.gemm_nt(dxb3[0, 1 + Nv:], st.wt("t2m"), dxt, M=Lq, batch=B, stride_a=Sm * H, stride_c=Lq * H, C2=dxtb)
        notify(st.slots[f"flava.text_model.encoder.layer.{self.nt - 1}.attention.attention.query.weight"].offset)
        # text stack
        for l in reversed(range(self.nt)):
            pfx = f"flava.text_model.encoder.layer.{l}."
            s = sv[f"t{l}"]
            dw = st.g(pfx + "attention.attention.adaptive_weight.0") if s["kw"]["sep"] is not None else None
            self._layer_bwd(pfx, f"t{l}", s, Mt, dxt, dxtb, dw=dw, want_f32=(l == 0))
            sv[f"t{l}"] = None
        s_t, tmean, trstd = sv["temb"]
        t = "flava.text_model.embeddings."
        dse = _e((Mt, H), F32, dev)
        ops.ln_bwd(dy_f32=dxt, s=s_t, mean=tmean, rstd=trstd, gamma=st.m(t + "LayerNorm.weight"), M=Mt, H=H, ds_f32=dse,
                   dgamma=st.g(t + "LayerNorm.weight"), dbeta=st.g(t + "LayerNorm.bias"))
        ops.text_embed_scatter(dse, sv["ids"], sv["tt"], st.g(t + "word_embeddings.weight"), st.g(t + "position_embeddings.weight"),
                               st.g(t + "token_type_embeddings.weight"), B, Lq, H)
        notify(st.slots[f"flava.image_model.encoder.layer.{self.ni - 1}.attention.attention.query.weight"].offset)
        # image stack + embeddings
        for l in reversed(range(self.ni)):
            self._layer_bwd(f"flava.image_model.encoder.layer.{l}.", f"i{l}", sv[f"i{l}"], Mi, dxi, dxib, want_f32=(l == 0))
            sv[f"i{l}"] = None
        e = "flava.image_model.embeddings."
        dpe = _e((B * 2 * P, H), BF, dev)
        ops.vision_assemble_bwd(dxi, dpe, st.g(e + "cls_token"), st.g(e + "position_embeddings"), B, P, H, tail_shift=1)
        gw = st.g(e + "patch_embeddings.projection.weight")
        ops.gemm_tn(dpe, sv["patches"], gw.view(H, -1), colsum=st.g(e + "patch_embeddings.projection.bias"))
        notify(st.total)
