"""fp32-accurate forward of the MKGformer path (evaluation / parity mode; see csrc/precise.hip).

Same schedule as ``engine.UnimoEngine.forward`` (UnimoEncoder.forward, MarT/models/modeling_unimo.py:589-658, and the
functions it calls), but every activation stays in fp32 and every dense contraction runs through the bf16 MFMA GEMM on
two-term operand splits (K' = 3K: hi*hi + lo*hi + hi*lo, fp32 accumulation), attention and the image-text fusion in
plain fp32 FMA arithmetic (``mart_attn_fwd_f32``).  LayerNorm, embeddings, GELU / quick-GELU epilogues, bias and residual
adds are the fp32 code of the fast path.  The reference computes in fp32; this mode is how the
``1e-3 on logits / exact ranked indices`` gate of BASELINE.json is met.  Forward only, dropout off.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import torch

from . import ops
from .params import FlatStore

BF, F32 = torch.bfloat16, torch.float32


def _e(shape, dtype, dev):
    return torch.empty(shape, device=dev, dtype=dtype)


class _PreciseBase:
    """Operand handling shared by the fp32-accurate forwards: cached weight splits, split-GEMM linear, fp32 LayerNorm, scoring."""
    st: FlatStore

    def _init_base(self, store: FlatStore):
        self.st = store
        self._w3: Dict[str, Tuple[int, torch.Tensor]] = {}
        # Error-budget hook (tools/error_budget.py): the components named here run with their operands ROUNDED TO BF16
        # (as the training path stores them) while everything else stays fp32-accurate, so the logit error each component class
        # contributes can be measured in isolation.  Tags: vis_lin, vis_attn, txt_lin_lo (layers < 8), txt_lin_hi, txt_attn,
        # fusion, head_t, head_s.  Empty in normal use.
        self.degrade: set = set()

    def _rb(self, x: torch.Tensor, tag: Optional[str]) -> torch.Tensor:
        return x.to(BF).to(F32) if tag is not None and tag in self.degrade else x


class PreciseUnimoForward(_PreciseBase):
    def __init__(self, store: FlatStore, vision_cfg, text_cfg):
        self._init_base(store)
        self.vc, self.tc = vision_cfg, text_cfg
        self.H, self.nh, self.I = text_cfg.hidden_size, text_cfg.num_attention_heads, text_cfg.intermediate_size
        self.n_layers = text_cfg.num_hidden_layers
        assert self.H // self.nh == 64
        self.eps_t, self.eps_v = float(text_cfg.layer_norm_eps), 1e-5
        self.fuse_from, self.export_from = 8, 7

    # ------------------------------------------------------------------ operands
    def w3(self, names: Sequence[str], rounded: bool = False) -> torch.Tensor:
        """[sum(out), 3*in] bf16 weight split (role 1) of one or several adjacent matrices, cached per store version."""
        key = names[0] + f"+{len(names)}" + ("~bf16" if rounded else "")
        ver = getattr(self.st, "version", 0)
        hit = self._w3.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        W = self.st.fused(list(names), self.st.master)
        W = W.reshape(W.shape[0], -1)
        if rounded:
            W = W.to(BF).to(F32)
        out = ops.split_bf16x3(W, 1)
        self._w3[key] = (ver, out)
        return out

    def lin(self, x: torch.Tensor, wnames: Sequence[str], bnames: Optional[Sequence[str]], N: int, tag: Optional[str] = None, **epi) -> torch.Tensor:
        """f32 [M, N] = epilogue(x @ W^T + b) with x f32 [M, K]."""
        out = _e((x.shape[0], N), F32, x.device)
        bias = self.st.fused(list(bnames), self.st.master) if bnames else None
        deg = tag is not None and tag in self.degrade
        ops.gemm_nt(ops.split_bf16x3(self._rb(x, tag), 0), self.w3(wnames, rounded=deg), out, bias=bias, **epi)
        return out

    def _ln(self, x, wname, bname, eps):
        M, H = x.shape
        y, mean, rstd = _e((M, H), F32, x.device), _e((M,), F32, x.device), _e((M,), F32, x.device)
        ops.ln_fwd(x_f32=x, gamma=self.st.m(wname), beta=self.st.m(bname), eps=eps, M=M, H=H, mean=mean, rstd=rstd, out_f32=y)
        return y

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, input_ids, attention_mask, token_type_ids, pixel_values, sep_idx, image_table=None, image_index=None):
        st, H, nh, I = self.st, self.H, self.nh, self.I
        dev = input_ids.device
        B, Lq = input_ids.shape
        S, p = self.vc.image_size, self.vc.patch_size
        P = (S // p) ** 2
        Nv = 1 + 2 * P
        Mv, Mt = B * Nv, B * Lq
        Kp = 3 * p * p
        # ---- vision embeddings (modeling_unimo.py:119-132,711)
        patches = _e((B * 2 * P, Kp), F32, dev)
        if image_index is not None:
            ops.patchify_f32(image_table, image_index.contiguous().view(-1), patches, B, S, p)
        else:
            ops.patchify_f32(pixel_values.contiguous(), None, patches, B, S, p)
        pe = self.lin(patches, ["unimo.vision_embeddings.patch_embedding.weight"], None, H)
        s_v = _e((Mv, H), F32, dev)
        ops.vision_assemble_f32(pe, st.m("unimo.vision_embeddings.class_embedding"), st.m("unimo.vision_embeddings.position_embedding.weight"),
                                s_v, B, P, H)
        xv = self._ln(s_v, "unimo.vision_pre_layrnorm.weight", "unimo.vision_pre_layrnorm.bias", self.eps_v)
        # ---- text embeddings (:152-186)
        u = "unimo.text_embeddings."
        s_t, tmean, trstd, xt = _e((Mt, H), F32, dev), _e((Mt,), F32, dev), _e((Mt,), F32, dev), _e((Mt, H), F32, dev)
        ops.text_embed_fwd(ids=input_ids, tt=token_type_ids, word=st.m(u + "word_embeddings.weight"), pos=st.m(u + "position_embeddings.weight"),
                           type_=st.m(u + "token_type_embeddings.weight"), gamma=st.m(u + "LayerNorm.weight"), beta=st.m(u + "LayerNorm.bias"),
                           eps=self.eps_t, p_drop=0.0, seed=0, B=B, Lq=Lq, H=H, s_out=s_t, mean=tmean, rstd=trstd, out_f32=xt, out_bf16=None)
        t_qkv_prev = None
        for l in range(self.n_layers):
            # ---- vision layer l (CLIPEncoderLayer.forward :490-527)
            v = f"unimo.encoder.vision_layers.{l}."
            h1 = self._ln(xv, v + "layer_norm1.weight", v + "layer_norm1.bias", self.eps_v)
            names = [v + f"self_attn.{n}" for n in ("q_proj", "k_proj", "v_proj")]
            qkv = self._rb(self.lin(h1, [n + ".weight" for n in names], [n + ".bias" for n in names], 3 * H, tag="vis_lin"), "vis_attn")
            ctx = _e((Mv, H), F32, dev)
            pre = t_qkv_prev if l >= self.fuse_from else None
            ops.attn_fwd_f32(q=qkv[:, :H], k=qkv[:, H:2 * H], v=qkv[:, 2 * H:], ctx=ctx, B=B, nh=nh, D=64, Sq=Nv, Sk=Nv, scale=0.125,
                             pk=pre[:, H:2 * H] if pre is not None else None, pv=pre[:, 2 * H:] if pre is not None else None,
                             Lp=Lq if pre is not None else 0)
            ctx = self._rb(ctx, "vis_attn")
            x1 = self.lin(ctx, [v + "self_attn.out_proj.weight"], [v + "self_attn.out_proj.bias"], H, tag="vis_lin", res_f32=xv)
            h2 = self._ln(x1, v + "layer_norm2.weight", v + "layer_norm2.bias", self.eps_v)
            f = self.lin(h2, [v + "mlp.fc1.weight"], [v + "mlp.fc1.bias"], I, tag="vis_lin", act=ops.ACT_QGELU)
            xv = self.lin(f, [v + "mlp.fc2.weight"], [v + "mlp.fc2.bias"], H, tag="vis_lin", res_f32=x1)
            # ---- text layer l (BertLayer.forward :540-577)
            t = f"unimo.encoder.text_layer.{l}."
            names = [t + f"attention.self.{n}" for n in ("query", "key", "value")]
            tl = "txt_lin_hi" if l >= self.fuse_from else "txt_lin_lo"
            tqkv = self._rb(self.lin(xt, [n + ".weight" for n in names], [n + ".bias" for n in names], 3 * H, tag=tl), "txt_attn")
            tctx = _e((Mt, H), F32, dev)
            on = sep_idx is not None
            ops.attn_fwd_f32(q=tqkv[:, :H], k=tqkv[:, H:2 * H], v=tqkv[:, 2 * H:], ctx=tctx, B=B, nh=nh, D=64, Sq=Lq, Sk=Lq, scale=0.125,
                             attn_mask=attention_mask, sep=sep_idx[:, 2:] if on else None, sep_stride=sep_idx.shape[1] if on else 0,
                             w0=st.m(t + "attention.self.adaptive_weight.0") if on else None,
                             w1=st.m(t + "attention.self.adaptive_weight.1") if on else None)
            tctx = self._rb(tctx, "txt_attn")
            fus = None
            if l >= self.fuse_from:                                  # BertFusion.forward :400-414 (unscaled, unmasked, one "head" of 768)
                fus = _e((Mt, H), F32, dev)
                xvf = self._rb(xv, "fusion")
                ops.attn_fwd_f32(q=self._rb(tctx, "fusion"), k=xvf, v=xvf, ctx=fus, B=B, nh=1, D=H, Sq=Lq, Sk=Nv, scale=1.0)
                fus = self._rb(fus, "fusion")
            s1 = self.lin(tctx, [t + "attention.output.dense.weight"], [t + "attention.output.dense.bias"], H, tag=tl, res_f32=xt)
            a = self._ln(s1, t + "attention.output.LayerNorm.weight", t + "attention.output.LayerNorm.bias", self.eps_t)
            ht = _e((Mt, I), F32, dev)
            deg = tl in self.degrade
            a3 = ops.split_bf16x3(self._rb(a, tl), 0)
            if fus is not None:
                ops.gemm_nt(a3, self.w3([t + "intermediate.dense.weight"], rounded=deg), ht, A2=ops.split_bf16x3(self._rb(fus, tl), 0),
                            B2=self.w3([t + "intermediate.fusion_dense.weight"], rounded=deg), bias=st.m(t + "intermediate.dense.bias"),
                            bias2=st.m(t + "intermediate.fusion_dense.bias"), act=ops.ACT_GELU)
            else:
                ops.gemm_nt(a3, self.w3([t + "intermediate.dense.weight"], rounded=deg), ht, bias=st.m(t + "intermediate.dense.bias"), act=ops.ACT_GELU)
            s2 = self.lin(ht, [t + "output.dense.weight"], [t + "output.dense.bias"], H, tag=tl, res_f32=a)
            xt = self._ln(s2, t + "output.LayerNorm.weight", t + "output.LayerNorm.bias", self.eps_t)
            t_qkv_prev = tqkv if l >= self.export_from else None
        # ---- MLM head transform (:972-975)
        hp = "cls.predictions.transform."
        y = self.lin(xt, [hp + "dense.weight"], [hp + "dense.bias"], H, tag="head_t", act=ops.ACT_GELU)
        trans = self._ln(y, hp + "LayerNorm.weight", hp + "LayerNorm.bias", self.eps_t)
        return trans.view(B, Lq, H)

    @torch.no_grad()
    def score(self, trans: torch.Tensor, rows: torch.Tensor, ids: torch.Tensor, word_name: str, bias_name: str) -> torch.Tensor:
        """logits[rows][:, ids] of the tied decoder (:958) on split operands."""
        H = trans.shape[-1]
        t3 = ops.split_bf16x3(self._rb(trans.reshape(-1, H), "head_s"), 0)
        out = _e((rows.numel(), ids.numel()), F32, trans.device)
        ops.gemm_nt(t3, self.w3([word_name], rounded="head_s" in self.degrade), out, a_rows=rows, b_rows=ids, bias=self.st.m(bias_name), bias_by_brow=True)
        return out


for _name in ("w3", "lin", "_ln", "score"):
    setattr(_PreciseBase, _name, PreciseUnimoForward.__dict__[_name])


class PreciseFlavaForward(_PreciseBase):
    """fp32-accurate twin of ``flava_engine.FlavaEngine.forward`` (MarT/models/modeling_flava.py:1373-1476, :2150-2204;
    SURVEY 8(a) row 19): three pre-LN stacks (image 1+P+P tokens with the tail-position quirk, text with the FLAVA variant
    of the adaptive reweight -- query row 0 exempt --, multimodal over 1 + image + text tokens without a mask), the two
    *_to_mm projections of the pre-final-layernorm streams, final multimodal layernorm, MLM head transform."""

    def __init__(self, store: FlatStore, config):
        self._init_base(store)
        self.cfg = config
        tc, ic, mc = config.text_config, config.image_config, config.multimodal_config
        self.H, self.nh, self.I = tc.hidden_size, tc.num_attention_heads, tc.intermediate_size
        assert self.H // self.nh == 64
        self.nt, self.ni, self.nm = tc.num_hidden_layers, ic.num_hidden_layers, mc.num_hidden_layers
        self.eps = float(tc.layer_norm_eps)

    def _block(self, p: str, x: torch.Tensor, B: int, S: int, **attn):
        """FlavaLayer.forward (:635-665): x + Attn(LN x); x + MLP(LN x)."""
        H, I = self.H, self.I
        a = p + "attention.attention."
        h1 = self._ln(x, p + "layernorm_before.weight", p + "layernorm_before.bias", self.eps)
        qkv = self.lin(h1, [a + f"{n}.weight" for n in ("query", "key", "value")], [a + f"{n}.bias" for n in ("query", "key", "value")], 3 * H)
        ctx = _e((x.shape[0], H), F32, x.device)
        ops.attn_fwd_f32(q=qkv[:, :H], k=qkv[:, H:2 * H], v=qkv[:, 2 * H:], ctx=ctx, B=B, nh=self.nh, D=64, Sq=S, Sk=S, scale=0.125, **attn)
        x1 = self.lin(ctx, [p + "attention.output.dense.weight"], [p + "attention.output.dense.bias"], H, res_f32=x)
        h2 = self._ln(x1, p + "layernorm_after.weight", p + "layernorm_after.bias", self.eps)
        f = self.lin(h2, [p + "intermediate.dense.weight"], [p + "intermediate.dense.bias"], I, act=ops.ACT_GELU)
        return self.lin(f, [p + "output.dense.weight"], [p + "output.dense.bias"], H, res_f32=x1)

    @torch.no_grad()
    def forward(self, input_ids, attention_mask, token_type_ids, pixel_values, sep_idx, image_table=None, image_index=None):
        st, H = self.st, self.H
        dev = input_ids.device
        B, Lq = input_ids.shape
        ic = self.cfg.image_config
        S, p = ic.image_size, ic.patch_size
        P = (S // p) ** 2
        Nv = 1 + 2 * P
        Sm = 1 + Nv + Lq
        Mi, Mt, Mm = B * Nv, B * Lq, B * Sm
        Kp = 3 * p * p
        # ---- image embeddings (:308-343) + stack
        patches = _e((B * 2 * P, Kp), F32, dev)
        if image_index is not None:
            ops.patchify_f32(image_table, image_index.contiguous().view(-1), patches, B, S, p)
        else:
            ops.patchify_f32(pixel_values.contiguous(), None, patches, B, S, p)
        e = "flava.image_model.embeddings."
        pe = self.lin(patches, [e + "patch_embeddings.projection.weight"], [e + "patch_embeddings.projection.bias"], H)
        xi = _e((Mi, H), F32, dev)
        ops.vision_assemble_f32(pe, st.m(e + "cls_token"), st.m(e + "position_embeddings"), xi, B, P, H, tail_shift=1)
        for l in range(self.ni):
            xi = self._block(f"flava.image_model.encoder.layer.{l}.", xi, B, Nv)
        # ---- text embeddings (:406-438) + stack (reweight variant :494-496)
        t = "flava.text_model.embeddings."
        s_t, tmean, trstd, xt = _e((Mt, H), F32, dev), _e((Mt,), F32, dev), _e((Mt,), F32, dev), _e((Mt, H), F32, dev)
        ops.text_embed_fwd(ids=input_ids, tt=token_type_ids, word=st.m(t + "word_embeddings.weight"), pos=st.m(t + "position_embeddings.weight"),
                           type_=st.m(t + "token_type_embeddings.weight"), gamma=st.m(t + "LayerNorm.weight"), beta=st.m(t + "LayerNorm.bias"),
                           eps=self.eps, p_drop=0.0, seed=0, B=B, Lq=Lq, H=H, s_out=s_t, mean=tmean, rstd=trstd, out_f32=xt, out_bf16=None)
        on = sep_idx is not None
        for l in range(self.nt):
            pfx = f"flava.text_model.encoder.layer.{l}."
            a = pfx + "attention.attention."
            xt = self._block(pfx, xt, B, Lq, attn_mask=attention_mask, sep=sep_idx[:, 2:] if on else None,
                             sep_stride=sep_idx.shape[1] if on else 0, w0=st.m(a + "adaptive_weight.0") if on else None,
                             w1=st.m(a + "adaptive_weight.1") if on else None, rw_skip_row0=True)
        # ---- multimodal input [cls | image_to_mm(img) | text_to_mm(txt)] (:1430,1450,1455-1456) + stack (no mask :1456)
        xm = _e((B, Sm, H), F32, dev)
        xm[:, 0, :].copy_(st.m("flava.multimodal_model.cls_token").view(1, H))
        ops.gemm_nt(ops.split_bf16x3(xi, 0), self.w3(["flava.image_to_mm_projection.weight"]), xm[0, 1:],
                    bias=st.m("flava.image_to_mm_projection.bias"), M=Nv, batch=B, stride_a=Nv * 3 * H, stride_c=Sm * H)
        ops.gemm_nt(ops.split_bf16x3(xt, 0), self.w3(["flava.text_to_mm_projection.weight"]), xm[0, 1 + Nv:],
                    bias=st.m("flava.text_to_mm_projection.bias"), M=Lq, batch=B, stride_a=Lq * 3 * H, stride_c=Sm * H)
        xm = xm.view(Mm, H)
        for l in range(self.nm):
            xm = self._block(f"flava.multimodal_model.encoder.layer.{l}.", xm, B, Sm)
        # ---- final layernorm, text positions, MLM head transform (:1209, :2187-2188, :1676-1680)
        mm = self._ln(xm, "flava.multimodal_model.layernorm.weight", "flava.multimodal_model.layernorm.bias", self.eps)
        rows = (torch.arange(B, device=dev, dtype=torch.int32)[:, None] * Sm + (1 + Nv) +
                torch.arange(Lq, device=dev, dtype=torch.int32)[None]).reshape(-1).contiguous()
        y = _e((Mt, H), F32, dev)
        ops.gemm_nt(ops.split_bf16x3(mm, 0), self.w3(["cls.transform.dense.weight"]), y, a_rows=rows, bias=st.m("cls.transform.dense.bias"),
                    act=ops.ACT_GELU)
        trans = self._ln(y, "cls.transform.LayerNorm.weight", "cls.transform.LayerNorm.bias", self.eps)
        return trans.view(B, Lq, H)
