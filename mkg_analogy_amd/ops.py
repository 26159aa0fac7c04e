"""Thin tensor-level wrappers over the C ABI (no autograd here; see engine.py / functional.py).

Every function enqueues HIP kernels on torch's current stream and returns immediately.  PyTorch is used for
device memory only.  bf16 tensors are torch.bfloat16; index tensors for gathers are int32.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _lib as L

ACT_NONE, ACT_GELU, ACT_QGELU, ACT_STORED = 0, 1, 2, 3      # ACT_STORED: mul_act only (mulz holds act'(z))
BF16, F32, F16 = torch.bfloat16, torch.float32, torch.float16


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def require_gpu() -> None:
    """Fail loudly when the HIP path cannot run (no silent CPU/eager fallback anywhere in the product)."""
    if not torch.cuda.is_available():
        raise L.MartError("mkg_analogy_amd needs a HIP device (MI355X / gfx950); there is no CPU fallback")
    L.check(L.lib().mart_check_device(), "mart_check_device")


def _rows2d(t: torch.Tensor):
    assert t.stride(-1) == 1, "innermost dimension must be contiguous"
    return t.stride(-2)


def gemm_nt(A, B, out, *, A2=None, B2=None, a_rows=None, b_rows=None, M=None, N=None, bias=None, bias2=None,
            bias_by_brow=False, act=ACT_NONE, preact=None, mulz=None, mul_act=ACT_NONE, res_f32=None, res_bf16=None,
            C2=None, alpha=1.0, batch=1, stride_a=0, stride_b=0, stride_c=0, stride_aux=0, tile_cfg=0, b_blocked=False,
            preact_grad=False, c_split3=False, row_stats=None, ln_mean=None, ln_rstd=None, ln_colsum=None):
    """out[M,N] = epi(A[M,K] @ B[N,K]^T (+ A2 @ B2^T)).  A/B are 2-D (row stride = ld) bf16 -- or fp16, all of them (the forward
    products of the text stream); out bf16 or f32, or fp16 with fp16 operands (C2 is then its bf16 copy).  ``c_split3``: out is bf16
    [M, 3N] and receives the f32 result as the two-term split [hi | lo | hi] (the A operand of the next GEMM of the fp32-accurate path)."""
    d = L.GemmNT()
    d.c_split3 = int(bool(c_split3))
    K = A.shape[-1]
    d.in_f16 = int(A.dtype == F16)
    assert B.dtype == A.dtype and (A2 is None or (A2.dtype == A.dtype and B2.dtype == A.dtype)), "gemm_nt operands must share one 16-bit type"
    d.c_f16 = int(out.dtype == F16)
    assert not d.c_f16 or d.in_f16, "an fp16 output needs fp16 operands"
    d.A, d.B, d.A2, d.B2 = _p(A), _p(B), _p(A2), _p(B2)
    d.lda, d.ldb = _rows2d(A), _rows2d(B)
    d.M = M if M is not None else (a_rows.numel() if a_rows is not None else A.shape[-2])
    d.N = N if N is not None else (b_rows.numel() if b_rows is not None else B.shape[-2])
    d.K, d.K2 = K, (A2.shape[-1] if A2 is not None else 0)
    if A2 is not None:
        assert _rows2d(A2) == d.lda and _rows2d(B2) == d.ldb
    d.a_rows, d.b_rows = _p(a_rows), _p(b_rows)
    d.a_src_rows = int(A.shape[-2]) if a_rows is not None else 0     # rows of the tables the gathers index (32-bit offset guard)
    d.b_src_rows = int(B.shape[-2]) if b_rows is not None else 0
    d.batch, d.stride_a, d.stride_b, d.stride_c, d.stride_aux = batch, stride_a, stride_b, stride_c, stride_aux
    d.bias, d.bias2, d.bias_by_brow = _p(bias), _p(bias2), int(bias_by_brow)
    d.act, d.preact, d.mulz, d.mul_act = act, _p(preact), _p(mulz), mul_act
    d.res_f32, d.res_bf16 = _p(res_f32), _p(res_bf16)
    res = res_f32 if res_f32 is not None else (res_bf16 if res_bf16 is not None else mulz)
    d.ldres = _rows2d(res) if res is not None else 0
    d.alpha = alpha
    d.C, d.ldc, d.c_f32 = _p(out), _rows2d(out), int(out.dtype == F32)
    d.C2, d.ldc2 = _p(C2), (_rows2d(C2) if C2 is not None else 0)
    d.tile_cfg = tile_cfg
    d.preact_grad = int(preact_grad)
    d.b_blocked = int(b_blocked)
    d.row_stats, d.ln_mean, d.ln_rstd, d.ln_colsum = _p(row_stats), _p(ln_mean), _p(ln_rstd), _p(ln_colsum)      # LayerNorm fold (mart_hip.h)
    L.check(L.lib().mart_gemm_nt(C.byref(d), _stream()), "mart_gemm_nt")
    return out


TN_DETERMINISTIC = os.environ.get("MART_DETERMINISTIC", "1") == "1"   # ordered (atomic-free) split reductions of gemm_tn and ln_bwd; 0 = f32 atomics


def gemm_tn(X, Y, out, *, M=None, NX=None, NY=None, out_rows=None, colsum=None, colsum_by_row=False, batch=1,
            stride_x=0, stride_y=0, stride_o=0, splits=0, alpha=1.0, deterministic=None):
    """out[NX,NY] (f32) += X[M,NX]^T @ Y[M,NY]; colsum[NX] += column sums of X.
    Deterministic by default (batch 1): the split-M partial tiles go through a workspace (allocated here from
    torch's caching allocator on the current stream) and are added in split order; otherwise f32 atomics."""
    d = L.GemmTN()
    d.X, d.Y, d.ldx, d.ldy = _p(X), _p(Y), _rows2d(X), _rows2d(Y)
    d.M = M if M is not None else X.shape[-2]
    d.NX = NX if NX is not None else X.shape[-1]
    d.NY = NY if NY is not None else Y.shape[-1]
    d.out, d.ldo, d.out_rows = _p(out), _rows2d(out), _p(out_rows)
    d.colsum, d.colsum_by_row = _p(colsum), int(colsum_by_row)
    d.batch, d.stride_x, d.stride_y, d.stride_o = batch, stride_x, stride_y, stride_o
    d.splits, d.alpha = splits, alpha
    det = TN_DETERMINISTIC if deterministic is None else deterministic
    ws = None
    if det and batch == 1:
        nb = int(L.lib().mart_gemm_tn_workspace_bytes(d.M, d.NX, d.NY, splits))
        ws = torch.empty(nb // 4, device=X.device, dtype=F32)
        d.workspace, d.workspace_bytes = _p(ws), nb
    L.check(L.lib().mart_gemm_tn(C.byref(d), _stream()), "mart_gemm_tn")
    return out


def ln_fwd(*, x_f32=None, y_bf16=None, y_f32=None, gamma, beta, eps, M, H, mean, rstd, s_out=None, out_f32=None, out_bf16=None,
           p_drop=0.0, seed=0, out_f16=None, x_rows=None, out_split3=None):
    d = L.LnFwd()
    d.out_f16, d.x_rows, d.out_split3 = _p(out_f16), _p(x_rows), _p(out_split3)
    d.x_f32, d.y_bf16, d.p_drop, d.seed = _p(x_f32), _p(y_bf16), p_drop, seed
    d.y_f32 = _p(y_f32)
    d.gamma, d.beta, d.eps, d.M, d.H = _p(gamma), _p(beta), eps, M, H
    d.s_out, d.out_f32, d.out_bf16, d.mean, d.rstd = _p(s_out), _p(out_f32), _p(out_bf16), _p(mean), _p(rstd)
    L.check(L.lib().mart_ln_fwd(C.byref(d), _stream()), "mart_ln_fwd")


def ln_bwd(*, dy_f32=None, dy_bf16=None, s, mean, rstd, gamma, M, H, add_f32=None, ds_f32=None, ds_bf16=None,
           p_drop=0.0, seed=0, dgamma=None, dbeta=None, bf16_total=False, add2_f32=None, defer_reduce=False, add_bf16=None):
    """LayerNorm backward.  ``defer_reduce`` (deterministic path only): the dgamma / dbeta partials stay in the returned workspace and the caller
    adds them with :func:`ln_dgb_reduce` (same kernel, same order) -- e.g. on the weight-gradient stream; returns ``(ws, partials)`` then."""
    d = L.LnBwd()
    d.add2_f32, d.add_bf16 = _p(add2_f32), _p(add_bf16)
    d.dy_f32, d.dy_bf16, d.s, d.mean, d.rstd, d.gamma = _p(dy_f32), _p(dy_bf16), _p(s), _p(mean), _p(rstd), _p(gamma)
    d.add_f32, d.M, d.H, d.ds_f32, d.ds_bf16 = _p(add_f32), M, H, _p(ds_f32), _p(ds_bf16)
    d.p_drop, d.seed, d.dgamma, d.dbeta = p_drop, seed, _p(dgamma), _p(dbeta)
    d.bf16_total = int(bf16_total)
    ws = None
    if TN_DETERMINISTIC and (dgamma is not None or dbeta is not None):     # ordered reduction of the per-workgroup dgamma / dbeta partials
        ws = torch.empty(768 * 2 * H, device=s.device, dtype=F32)
        d.ws, d.ws_bytes = _p(ws), ws.numel() * 4
    defer = bool(defer_reduce) and ws is not None
    d.defer_reduce = int(defer)
    L.check(L.lib().mart_ln_bwd(C.byref(d), _stream()), "mart_ln_bwd")
    return (ws, int(L.lib().mart_ln_bwd_partials(M))) if defer else None


def ln_fold_prep(W, bias, gamma, beta, Wf, s, bf):
    """Operands of the LayerNorm fold: Wf = bf16(gamma o W), s[n] = sum_k Wf[n, k], bf = bias + W beta (f32 W [N, K])."""
    N, K = W.shape
    L.check(L.lib().mart_ln_fold_prep(_p(W), _p(bias), _p(gamma), _p(beta), _p(Wf), _p(s), _p(bf), N, K, _stream()), "mart_ln_fold_prep")


def ln_stats_finalize(partials, M, H, eps, mean, rstd):
    L.check(L.lib().mart_ln_stats_finalize(_p(partials), M, H, eps, _p(mean), _p(rstd), _stream()), "mart_ln_stats_finalize")


def ln_dgb_reduce(ws, partials, H, dgamma, dbeta):
    L.check(L.lib().mart_ln_dgb_reduce(_p(ws), partials, H, _p(dgamma), _p(dbeta), _stream()), "mart_ln_dgb_reduce")


def patchify(pixels, out, B, S, p):
    L.check(L.lib().mart_patchify(_p(pixels), _p(out), B, S, p, _stream()), "mart_patchify")


def patchify_gather(table, index, out, B, S, p):
    L.check(L.lib().mart_patchify_gather(_p(table), _p(index), _p(out), B, S, p, _stream()), "mart_patchify_gather")


def gather_images(table, index, out, B, S):
    L.check(L.lib().mart_gather_images(_p(table), _p(index), _p(out), B, S, _stream()), "mart_gather_images")


def vision_assemble(patch, cls, pos, s, B, P, H, tail_shift=0):
    L.check(L.lib().mart_vision_assemble(_p(patch), _p(cls), _p(pos), _p(s), B, P, H, tail_shift, _stream()), "mart_vision_assemble")


def vision_assemble_bwd(ds, dpatch, dcls, dpos, B, P, H, tail_shift=0):
    if TN_DETERMINISTIC:                               # slice partials + ordered reduction instead of f32 atomics
        ws = torch.empty((P + 1) * 16 * 2 * H, device=ds.device, dtype=F32)
        L.check(L.lib().mart_vision_assemble_bwd_det(_p(ds), _p(dpatch), _p(dcls), _p(dpos), B, P, H, tail_shift, _p(ws), ws.numel() * 4, _stream()),
                "mart_vision_assemble_bwd_det")
        return
    L.check(L.lib().mart_vision_assemble_bwd(_p(ds), _p(dpatch), _p(dcls), _p(dpos), B, P, H, tail_shift, _stream()), "mart_vision_assemble_bwd")


def text_embed_fwd(*, ids, tt, word, pos, type_, gamma, beta, eps, p_drop, seed, B, Lq, H, s_out, mean, rstd, out_f32, out_bf16, out_f16=None):
    d = L.TextEmbed()
    d.out_f16 = _p(out_f16)
    d.ids, d.tt, d.word, d.pos, d.type = _p(ids), _p(tt), _p(word), _p(pos), _p(type_)
    d.gamma, d.beta, d.eps, d.p_drop, d.seed = _p(gamma), _p(beta), eps, p_drop, seed
    d.B, d.L, d.H = B, Lq, H
    d.s_out, d.mean, d.rstd, d.out_f32, d.out_bf16 = _p(s_out), _p(mean), _p(rstd), _p(out_f32), _p(out_bf16)
    L.check(L.lib().mart_text_embed_fwd(C.byref(d), _stream()), "mart_text_embed_fwd")


def dropout_bwd_f32(dy_f32, dy_bf16, out, n, p, seed):
    L.check(L.lib().mart_dropout_bwd_f32(_p(dy_f32), _p(dy_bf16), _p(out), n, p, seed, _stream()), "mart_dropout_bwd_f32")


def text_embed_scatter(ds, ids, tt, dword, dpos, dtype, B, Lq, H):
    if TN_DETERMINISTIC:                               # word rows summed in sorted-token order, position / type rows through slice partials
        order = torch.argsort(ids.reshape(-1), stable=True)          # index preparation only; the sums are the kernels'
        nwin = (B * Lq + 63) // 64
        ws = torch.empty((nwin * 2 + Lq * 16 * 2) * H, device=ds.device, dtype=F32)
        meta = torch.empty(nwin, device=ds.device, dtype=torch.int32)
        L.check(L.lib().mart_text_embed_scatter_det(_p(ds), _p(ids), _p(tt), _p(order), _p(dword), _p(dpos), _p(dtype), B, Lq, H,
                                                    _p(ws), ws.numel() * 4, _p(meta), _stream()), "mart_text_embed_scatter_det")
        return
    L.check(L.lib().mart_text_embed_scatter(_p(ds), _p(ids), _p(tt), _p(dword), _p(dpos), _p(dtype), B, Lq, H, _stream()),
            "mart_text_embed_scatter")


def _attn_desc(d, *, q, k, v, ctx, lse, B, nh, Sq, Sk, scale, pk=None, pv=None, Lp=0, attn_mask=None, sep=None,
               sep_stride=0, w0=None, w1=None, p_drop=0.0, seed=0, rw_skip_row0=False, ctx_f16=None):
    d.q, d.k, d.v = _p(q), _p(k), _p(v)
    d.ctx_f16 = _p(ctx_f16)
    assert ctx_f16 is None or (_rows2d(ctx_f16) == _rows2d(ctx) and ctx_f16.dtype == F16)
    d.ldq, d.ldk, d.ldv = _rows2d(q), _rows2d(k), _rows2d(v)
    d.pk, d.pv, d.ldp, d.Lp = _p(pk), _p(pv), (_rows2d(pk) if pk is not None else 0), Lp
    d.B, d.nh, d.Sq, d.Sk, d.scale = B, nh, Sq, Sk, scale
    d.attn_mask, d.sep, d.sep_stride, d.w0, d.w1 = _p(attn_mask), _p(sep), sep_stride, _p(w0), _p(w1)
    d.p_drop, d.seed = p_drop, seed
    d.ctx, d.ldctx, d.lse = _p(ctx), _rows2d(ctx), _p(lse)
    d.rw_skip_row0 = int(rw_skip_row0)


def attn_fwd(**kw):
    d = L.AttnFwd()
    _attn_desc(d, **kw)
    L.check(L.lib().mart_attn_fwd(C.byref(d), _stream()), "mart_attn_fwd")


def attn_bwd(*, dctx, delta, dq, dk, dv, dpk=None, dpv=None, accum_dkv=False, dw=None, **kw):
    d = L.AttnBwd()
    _attn_desc(d.f, **kw)
    d.dctx, d.lddctx, d.delta = _p(dctx), _rows2d(dctx), _p(delta)
    d.dq, d.dk, d.dv = _p(dq), _p(dk), _p(dv)
    d.lddq, d.lddk, d.lddv = _rows2d(dq), _rows2d(dk), _rows2d(dv)
    d.dpk, d.dpv, d.lddp = _p(dpk), _p(dpv), (_rows2d(dpk) if dpk is not None else 0)
    d.accum_dkv, d.dw = int(accum_dkv), _p(dw)
    ws = None
    if dw is not None:                               # per-wave partials of d(w0), d(w1) instead of contended atomics
        ws = torch.empty(2 * 4 * kw["B"] * kw["nh"] * ((kw["Sq"] + 127) // 128), device=dw.device, dtype=F32)
        d.dw_ws = _p(ws)
    L.check(L.lib().mart_attn_bwd(C.byref(d), _stream()), "mart_attn_bwd")


# ---------------------------------------------------------------- fp32-accurate evaluation path (csrc/precise.hip)
def split_bf16x3(src, role, out=None, terms=2):
    """f32 [rows, K] (row stride = ld) -> bf16 [rows, 3K]: role 0 = [hi|lo|hi] (A operand), role 1 = [hi|hi|lo] (B operand).
    ``terms=3``: three-term splits for six products, [rows, 6K] (csrc/precise.hip)."""
    rows, K = src.shape
    if out is None:
        out = torch.empty((rows, (6 if terms == 3 else 3) * K), device=src.device, dtype=BF16)
    L.check(L.lib().mart_split_bf16x3(_p(src), _rows2d(src), _p(out), rows, K, role, terms, _stream()), "mart_split_bf16x3")
    return out


def split_bf16x3_rows(src, gather, role, out=None, terms=2):
    """Same split of the gathered rows ``src[gather]`` (gather int32 [R]) -> bf16 [R, 3K] (or [R, 6K])."""
    K = src.shape[1]
    R = gather.numel()
    assert gather.dtype == torch.int32 and gather.is_contiguous()
    if out is None:
        out = torch.empty((R, (6 if terms == 3 else 3) * K), device=src.device, dtype=BF16)
    L.check(L.lib().mart_split_bf16x3_rows(_p(src), _rows2d(src), _p(gather), _p(out), R, K, role, terms, _stream()), "mart_split_bf16x3_rows")
    return out


def patchify_f32(pixels_or_table, index, out, B, S, p):
    L.check(L.lib().mart_patchify_f32(_p(pixels_or_table), _p(index), _p(out), B, S, p, _stream()), "mart_patchify_f32")


def vision_assemble_f32(patch, cls, pos, s, B, P, H, tail_shift=0):
    L.check(L.lib().mart_vision_assemble_f32(_p(patch), _p(cls), _p(pos), _p(s), B, P, H, tail_shift, _stream()), "mart_vision_assemble_f32")


def attn_fwd_f32(*, q, k, v, ctx, B, nh, D, Sq, Sk, scale, pk=None, pv=None, Lp=0, attn_mask=None, sep=None, sep_stride=0,
                 w0=None, w1=None, rw_skip_row0=False, fast=False, ctx_split3=None):
    """``fast`` (evaluation passes): unmasked head-dim-64 calls may run on two-term bf16 operand splits (csrc/attention.hip attn_split_fwd_k);
    ``ctx_split3`` (fast path only, bf16 [rows, 3 * nh * D]): the context as the [hi | lo | hi] operand of the output projection (``ctx`` may be None)."""
    d = L.AttnF32()
    d.fast = int(bool(fast))
    d.ctx_split3, d.ldctx3 = _p(ctx_split3), (_rows2d(ctx_split3) if ctx_split3 is not None else 0)
    d.q, d.k, d.v, d.ldq, d.ldk, d.ldv = _p(q), _p(k), _p(v), _rows2d(q), _rows2d(k), _rows2d(v)
    d.pk, d.pv, d.ldp, d.Lp = _p(pk), _p(pv), (_rows2d(pk) if pk is not None else 0), Lp
    d.B, d.nh, d.D, d.Sq, d.Sk, d.scale = B, nh, D, Sq, Sk, scale
    d.attn_mask, d.sep, d.sep_stride, d.w0, d.w1, d.rw_skip_row0 = _p(attn_mask), _p(sep), sep_stride, _p(w0), _p(w1), int(rw_skip_row0)
    d.ctx, d.ldctx = _p(ctx), (_rows2d(ctx) if ctx is not None else 0)
    L.check(L.lib().mart_attn_fwd_f32(C.byref(d), _stream()), "mart_attn_fwd_f32")


def attn_bwd_f32(*, dctx, dq, dk, dv, dpk=None, dpv=None, dw=None, q, k, v, B, nh, D, Sq, Sk, scale, pk=None, pv=None, Lp=0, attn_mask=None,
                 sep=None, sep_stride=0, w0=None, w1=None, rw_skip_row0=False):
    """fp32 attention backward (verification mode): dq written; dk / dv / dpk / dpv / dw accumulated -- zero them first."""
    d = L.AttnBwdF32()
    f = d.f
    f.q, f.k, f.v, f.ldq, f.ldk, f.ldv = _p(q), _p(k), _p(v), _rows2d(q), _rows2d(k), _rows2d(v)
    f.pk, f.pv, f.ldp, f.Lp = _p(pk), _p(pv), (_rows2d(pk) if pk is not None else 0), Lp
    f.B, f.nh, f.D, f.Sq, f.Sk, f.scale = B, nh, D, Sq, Sk, scale
    f.attn_mask, f.sep, f.sep_stride, f.w0, f.w1, f.rw_skip_row0 = _p(attn_mask), _p(sep), sep_stride, _p(w0), _p(w1), int(rw_skip_row0)
    f.ctx, f.ldctx = None, 0
    d.dctx, d.lddctx = _p(dctx), _rows2d(dctx)
    d.dq, d.dk, d.dv, d.lddq, d.lddk, d.lddv = _p(dq), _p(dk), _p(dv), _rows2d(dq), _rows2d(dk), _rows2d(dv)
    d.dpk, d.dpv, d.lddp = _p(dpk), _p(dpv), (_rows2d(dpk) if dpk is not None else 0)
    d.dw = _p(dw)
    L.check(L.lib().mart_attn_bwd_f32(C.byref(d), _stream()), "mart_attn_bwd_f32")


def split_bf16x3_stack(src, role, terms=2):
    """f32 [M, K] -> bf16 [3M, K] row-stacked two-term split: role 0 = [hi;lo;hi] (X of X^T Y), role 1 = [hi;hi;lo] (Y); terms=3: [6M, K]."""
    M, K = src.shape
    out = torch.empty(((6 if terms == 3 else 3) * M, K), device=src.device, dtype=BF16)
    L.check(L.lib().mart_split_bf16x3_stack(_p(src), _rows2d(src), _p(out), M, K, role, terms, _stream()), "mart_split_bf16x3_stack")
    return out


def act_f32(z, act):
    a = torch.empty_like(z)
    L.check(L.lib().mart_act_f32(_p(z), _p(a), act, z.numel(), _stream()), "mart_act_f32")
    return a


def act_bwd_f32(dy, z, act):
    dz = torch.empty_like(z)
    L.check(L.lib().mart_act_bwd_f32(_p(dy), _p(z), act, _p(dz), z.numel(), _stream()), "mart_act_bwd_f32")
    return dz


def colsum_f32(src, out):
    """out[C] += column sums of the f32 matrix ``src`` [R, C] (row stride = ld)."""
    R, Cc = src.shape
    L.check(L.lib().mart_colsum_f32(_p(src), _rows2d(src), _p(out), R, Cc, _stream()), "mart_colsum_f32")


def fusion_supported(Lq, Nv, H) -> bool:
    return bool(L.lib().mart_fusion_supported(int(Lq), int(Nv), int(H)))


def fusion_fwd(q, v, out, probs, B, Lq, Nv, H, out_f16=None):
    """BertFusion forward in one kernel (modeling_unimo.py:400-414): out = softmax(q v^T) v, probs saved for the backward pass."""
    d = L.FusionFwd()
    d.out_f16 = _p(out_f16)
    assert out_f16 is None or (_rows2d(out_f16) == _rows2d(out) and out_f16.dtype == F16)
    d.q, d.ldq, d.v, d.ldv, d.out, d.ldo, d.probs, d.ldp = _p(q), _rows2d(q), _p(v), _rows2d(v), _p(out), _rows2d(out), _p(probs), _rows2d(probs)
    d.B, d.Lq, d.Nv, d.H = B, Lq, Nv, H
    L.check(L.lib().mart_fusion_fwd(C.byref(d), _stream()), "mart_fusion_fwd")


def fusion_bwd(q, v, dout, probs, dq, dv_f32, dv_bf16, B, Lq, Nv, H):
    d = L.FusionBwd()
    d.q, d.ldq, d.v, d.ldv, d.dout, d.lddo, d.probs, d.ldp = _p(q), _rows2d(q), _p(v), _rows2d(v), _p(dout), _rows2d(dout), _p(probs), _rows2d(probs)
    d.dq, d.lddq, d.dv_f32, d.lddv = _p(dq), _rows2d(dq), _p(dv_f32), (_rows2d(dv_f32) if dv_f32 is not None else 0)
    d.dv_bf16, d.lddvb = _p(dv_bf16), (_rows2d(dv_bf16) if dv_bf16 is not None else 0)
    d.B, d.Lq, d.Nv, d.H = B, Lq, Nv, H
    L.check(L.lib().mart_fusion_bwd(C.byref(d), _stream()), "mart_fusion_bwd")


def softmax_fwd(scores, probs, R, Cc):
    L.check(L.lib().mart_softmax_fwd(_p(scores), _rows2d(scores), _p(probs), _rows2d(probs), R, Cc, _stream()), "mart_softmax_fwd")


def softmax_bwd(probs, dprobs, dscores, R, Cc):
    L.check(L.lib().mart_softmax_bwd(_p(probs), _rows2d(probs), _p(dprobs), _rows2d(dprobs), _p(dscores), _rows2d(dscores), R, Cc,
                                     _stream()), "mart_softmax_bwd")


def transpose_bf16(inp, out, R, Cc, Rp, batch=1, stride_i=0, stride_o=0):
    L.check(L.lib().mart_transpose_bf16(_p(inp), _rows2d(inp), stride_i, _p(out), Rp, stride_o, R, Cc, batch, _stream()),
            "mart_transpose_bf16")


REDUCE = {"none": 0, "mean": 1, "sum": 2}


def lsce_fwd(logits, label, eps, loss_rows, lse, ignore_index=-100, loss_out=None, reduction="none", status=None):
    """``loss_out`` (f32 [2] on the device): {reduced loss, n_valid}; ``status`` (int32 [1]): set to 1 by a label outside [0, C) u {ignore}."""
    R, Cc = logits.shape
    L.check(L.lib().mart_lsce_fwd(_p(logits), _rows2d(logits), _p(label), int(ignore_index), eps, _p(loss_rows), _p(lse), _p(loss_out), REDUCE[reduction],
                                  _p(status), R, Cc, _stream()), "mart_lsce_fwd")


def lsce_bwd(logits, label, lse, eps, gscale, rowscale, dl_bf16=None, dl_f32=None, ignore_index=-100, n_valid=None, gscale_per_row=False):
    R, Cc = logits.shape
    ldo = _rows2d(dl_bf16) if dl_bf16 is not None else Cc
    L.check(L.lib().mart_lsce_bwd(_p(logits), _rows2d(logits), _p(label), int(ignore_index), _p(lse), eps, _p(gscale), int(gscale_per_row), rowscale,
                                  _p(n_valid), _p(dl_bf16), ldo, _p(dl_f32), R, Cc, _stream()), "mart_lsce_bwd")


def rank(logits, label, out):
    R, Cc = logits.shape
    L.check(L.lib().mart_rank(_p(logits), _rows2d(logits), _p(label), _p(out), R, Cc, _stream()), "mart_rank")


def simloss_fwd(trans, rel_idx, q_idx, a_idx, loss_rows, rows=None, L_=None):
    """``rows`` (int32 [B, nr] flat ids) given: ``trans`` is the compact [B * nr, H] tensor of a row-subset pass and ``L_`` the sequence length."""
    if rows is None:
        B, Lq, H = trans.shape
        nr = 0
    else:
        B, nr = rows.shape
        Lq, H = int(L_), trans.shape[-1]
    L.check(L.lib().mart_simloss_fwd(_p(trans), _p(rel_idx), _p(q_idx), _p(a_idx), _p(rows), nr, _p(loss_rows), B, Lq, H, _stream()), "mart_simloss_fwd")


def simloss_bwd(trans, rel_idx, q_idx, a_idx, gscale, rowscale, dtrans, rows=None, L_=None):
    if rows is None:
        B, Lq, H = trans.shape
        nr = 0
    else:
        B, nr = rows.shape
        Lq, H = int(L_), trans.shape[-1]
    L.check(L.lib().mart_simloss_bwd(_p(trans), _p(rel_idx), _p(q_idx), _p(a_idx), _p(rows), nr, _p(gscale), rowscale, _p(dtrans), B, Lq, H, _stream()),
            "mart_simloss_bwd")


def needed_rows(ids, token, rel_idx, q_idx, a_idx, rows_out, mask_row_out=None, status=None):
    B, Lq = ids.shape
    L.check(L.lib().mart_needed_rows(_p(ids), B, Lq, int(token), _p(rel_idx), _p(q_idx), _p(a_idx), _p(rows_out), _p(mask_row_out), _p(status), _stream()),
            "mart_needed_rows")


def rows_lookup(flat, rows, Lq, out, status=None):
    L.check(L.lib().mart_rows_lookup(_p(flat), flat.numel(), _p(rows), int(rows.shape[1]), int(Lq), _p(out), _p(status), _stream()), "mart_rows_lookup")


def sum_splits_f32(parts, out):
    L.check(L.lib().mart_sum_splits_f32(_p(parts), int(parts.shape[0]), out.numel(), _p(out), _stream()), "mart_sum_splits_f32")


def rows_dense(src, rows, B, Lq, dst, fill=float("nan")):
    L.check(L.lib().mart_rows_dense(_p(src), _p(rows), int(rows.shape[1]), B, Lq, src.shape[-1], _p(dst), fill, _stream()), "mart_rows_dense")


def find_token(ids, token, pos_out, row_out=None, status=None):
    B, Lq = ids.shape
    L.check(L.lib().mart_find_token(_p(ids), B, Lq, token, _p(pos_out), _p(row_out), _p(status), _stream()), "mart_find_token")


def cast_f32_bf16(src, dst):
    L.check(L.lib().mart_cast_f32_bf16(_p(src), _p(dst), src.numel(), _stream()), "mart_cast_f32_bf16")


def cast_bf16_f32(src, dst):
    L.check(L.lib().mart_cast_bf16_f32(_p(src), _p(dst), src.numel(), _stream()), "mart_cast_bf16_f32")


def cast_f32_f16(src, dst):
    L.check(L.lib().mart_cast_f32_f16(_p(src), _p(dst), src.numel(), _stream()), "mart_cast_f32_f16")


def cast_bf16_f16(src, dst):
    L.check(L.lib().mart_cast_bf16_f16(_p(src), _p(dst), src.numel(), _stream()), "mart_cast_bf16_f16")


def gather_rows_first_f32(src, rows, group, dst):
    """dst[r] = src[rows[r]], or 0 where an earlier slot of the same group (``group`` consecutive entries = one example) names the same row."""
    R, H = dst.shape
    L.check(L.lib().mart_gather_rows_first_f32(_p(src), _rows2d(src), _p(rows), group, _p(dst), R, H, _stream()), "mart_gather_rows_first_f32")


def scatter_rows(src, rows, group, dst, accumulate=False):
    """dst[rows[r]] (+)= src[r] (src f32 [R,H]; dst f32 or bf16), the slots of a group applied in order by one workgroup (repeats allowed)."""
    R, H = src.shape
    assert src.dtype == F32 and dst.dtype in (F32, BF16)
    L.check(L.lib().mart_scatter_rows(_p(src), _p(rows), group, _p(dst), _rows2d(dst), int(dst.dtype == BF16), int(accumulate), R, H, _stream()),
            "mart_scatter_rows")


def cast_pad_f32_bf16(src, dst, R, Cc):
    L.check(L.lib().mart_cast_pad_f32_bf16(_p(src), _rows2d(src), _p(dst), _rows2d(dst), R, Cc, _stream()), "mart_cast_pad_f32_bf16")


def gather_rows_bf16(src, rows, dst):
    R, H = dst.shape
    L.check(L.lib().mart_gather_rows_bf16(_p(src), _rows2d(src), _p(rows), _p(dst), R, H, _stream()), "mart_gather_rows_bf16")


def act_bwd(dy, z, act, out):
    L.check(L.lib().mart_act_bwd(_p(dy), _p(z), act, _p(out), dy.numel(), _stream()), "mart_act_bwd")


def gather_rows_f32(src, rows, dst):
    R, H = dst.shape
    L.check(L.lib().mart_gather_rows_f32(_p(src), _rows2d(src), _p(rows), _p(dst), R, H, _stream()), "mart_gather_rows_f32")


def scatter_add_rows_f32(src, rows, dst):
    R, H = src.shape
    L.check(L.lib().mart_scatter_add_rows_f32(_p(src), _p(rows), _p(dst), _rows2d(dst), R, H, _stream()), "mart_scatter_add_rows_f32")


def add_f32_bf16(a, b_bf16=None, out_f32=None, out_bf16=None):
    L.check(L.lib().mart_add_f32_bf16(_p(a), _p(b_bf16), _p(out_f32), _p(out_bf16), a.numel(), _stream()), "mart_add_f32_bf16")


def dropout_mask(out_u8, p, seed):
    L.check(L.lib().mart_dropout_mask(_p(out_u8), out_u8.numel(), p, seed, _stream()), "mart_dropout_mask")


def adamw(*, master, grad, m, v, shadow, chunks, n_chunks, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale=1.0, shadow_f16=None, zero_grad=False):
    d = L.AdamW()
    d.shadow_f16 = _p(shadow_f16)
    d.zero_grad = int(zero_grad)
    d.master, d.grad, d.m, d.v, d.shadow_bf16 = _p(master), _p(grad), _p(m), _p(v), _p(shadow)
    d.chunks, d.n_chunks = _p(chunks), n_chunks
    d.lr, d.beta1, d.beta2, d.eps, d.weight_decay, d.bc1, d.bc2, d.grad_scale = lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale
    L.check(L.lib().mart_adamw(C.byref(d), _stream()), "mart_adamw")


def block_table(src, dst, table, n):
    L.check(L.lib().mart_block_table(_p(src), _p(dst), _p(table), n, _stream()), "mart_block_table")


def transpose_table(src, dst, table, n):
    L.check(L.lib().mart_transpose_table(_p(src), _p(dst), _p(table), n, _stream()), "mart_transpose_table")
