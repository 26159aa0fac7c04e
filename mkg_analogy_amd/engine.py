"""Hand-written forward/backward of the MKGformer ("Unimo") encoder + MLM head transform on the HIP kernels.

This is the hot loop of the reference (MarT/models/modeling_unimo.py:589-658 UnimoEncoder.forward and everything
it calls) re-scheduled for MI355X: bf16 MFMA contractions with fused epilogues, flash attention with the text K/V
prefix read in place, fp32 residual streams / LayerNorm statistics / softmax, activations saved once for a manual
backward that accumulates straight into the flat fp32 gradient buffer (params.FlatStore).  PyTorch supplies device
memory and the outer autograd node (functional.py) only.

Cross-wiring reproduced exactly (modeling_unimo.py:616,627-628): vision layer idx prepends the (K,V) of text layer
idx-1 as key/value prefix iff idx>=8; text layer idx fuses with the output of vision layer idx iff idx>=8.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import collections
import os

import torch

from . import ops
from .params import FlatStore

BF, F32, HF = torch.bfloat16, torch.float32, torch.float16


def _e(shape, dtype, dev):
    return torch.empty(shape, device=dev, dtype=dtype)


class UnimoEngine:
    def __init__(self, store: FlatStore, vision_cfg, text_cfg):
        self.st = store
        self.vc, self.tc = vision_cfg, text_cfg
        self.H = text_cfg.hidden_size
        self.nh = text_cfg.num_attention_heads
        self.I = text_cfg.intermediate_size
        self.n_layers = text_cfg.num_hidden_layers
        assert vision_cfg.hidden_size == self.H and vision_cfg.num_attention_heads == self.nh, \
            "the HIP path is built for equal text/vision widths (BERT-base + CLIP ViT-B)"
        assert self.H // self.nh == 64, "attention kernels are specialised for head_dim 64"
        assert vision_cfg.num_hidden_layers == self.n_layers
        self.eps_t = float(text_cfg.layer_norm_eps)
        self.eps_v = 1e-5                      # nn.LayerNorm default: modeling_unimo.py:486-488,682 ignore the config eps
        self.p_hidden = float(text_cfg.hidden_dropout_prob)
        self.p_attn = float(text_cfg.attention_probs_dropout_prob)
        self.fuse_from = 8                     # modeling_unimo.py:616,627
        self.export_from = 7                   # modeling_unimo.py:628
        self.grad_ready: Optional[Callable[[int], None]] = None      # DDP hook: gradients below this flat offset are final
        self.grad_ready_async: Optional[Callable] = None              # same, without joining the streams: called with (offset, events to wait for)
        self.taps: Optional[dict] = None                              # debugging: per-layer stream snapshots when set to a dict
        # Teacher forcing (tests only, single-stream schedule): ``inject[name]`` replaces the stream that ``taps[name]`` would have
        # recorded ("vis_emb", "txt_emb", "vis{l}", "txt{l}": the INPUT of the next layer), ``inject_grad["vis{l}" / "txt{l}"]``
        # replaces the gradient w.r.t. that stream in the backward pass (the engine's own value is tapped as "dvis{l}" / "dtxt{l}"
        # first).  Every layer then sees exact inputs and exact upstream gradients, so a per-layer error is that layer's own.
        self.inject: Optional[dict] = None
        self.inject_grad: Optional[dict] = None
        import os
        self.ln_defer = os.environ.get("MART_LN_DEFER", "0") == "1"            # LayerNorm dgamma / dbeta reductions on the weight-gradient stream (measured: +0.45 ms, off)
        self.overlap_wgrad = os.environ.get("MART_OVERLAP_WGRAD", "1") == "1"   # weight-gradient GEMMs on a side stream (+2.5 % step rate)
        self._side: Optional[torch.cuda.Stream] = None
        self._side_busy = False
        self._side_t: Optional[torch.cuda.Stream] = None               # second weight-gradient queue (launches issued from the text stream), wgrad_two
        self._side_t_busy = False
        # wgrad_two: a second weight-gradient queue for the launches the TEXT stream issues (one in-order queue makes a vision weight-gradient GEMM whose
        # operands are ready wait behind a text one whose event the text stream has not reached).  Pays where the text stream is a large part of the
        # step -- 49 patches: 34.63 -> 33.58 ms (-3.0 %, three alternations) -- and costs at 196 patches (84.7 -> 85.3 ms: three queues of big GEMMs
        # interleave worse than two), so "auto" (default) turns it on when the text rows are at least a quarter of the vision rows.  MART_WGRAD_TWO=0|1 forces.
        self.wgrad_two = os.environ.get("MART_WGRAD_TWO", "auto")
        self._wg2 = False
        self.fused_fusion = os.environ.get("MART_FUSION_FUSED", "1") == "1"   # BertFusion as one kernel per direction where the shape allows
        self.two_stream = os.environ.get("MART_TWO_STREAM", "1") == "1"   # text layers on their own stream (+3 % step rate)
        self._tstream: Optional[torch.cuda.Stream] = None
        # Host run-ahead bound.  Blocks that side-stream kernels touched (record_stream) return to the caching allocator only
        # when the recorded stream has passed the point of the free; a host that enqueues many steps ahead of the GPU finds
        # none of them reusable and reserves ~2.3 GiB more per queued step (B=256) until the 288 GB are gone and every
        # allocation turns into free-everything-and-retry (seen as 100 -> 600 ms per step after ~60 un-synchronised steps).
        # The pass that is about to be enqueued waits for the end of the pass two before it: the GPU queue never drains.
        self.max_inflight = int(os.environ.get("MART_MAX_INFLIGHT", "2"))
        self._inflight = collections.deque()
        # Where the bf16 logit error comes from (tools/error_budget.py, conditioned weights): head transform 20 % + scoring 16 % of the
        # variance for 0.01 % of the FLOPs.  With head_split the transform GEMM (and the scoring GEMM, functional._ScoreFn) runs on
        # two-term bf16 operand splits of the f32 text stream / f32 master weights (K' = 3K: hi*hi + lo*hi + hi*lo, csrc/precise.hip);
        # the backward pass is unchanged (bf16 operands).
        self.head_split = os.environ.get("MART_HEAD_SPLIT", "1") == "1"
        # text_f16: the FORWARD linear layers of the text stream (16 k of the 117 k rows of a step, 14 % of its FLOPs, but 58 % of the bf16 logit
        # error variance on well-conditioned weights: tools/error_budget.py) multiply fp16 operands instead of bf16 ones -- the same MFMA rate
        # (v_mfma_f32_32x32x16_f16), 11-bit instead of 8-bit significands, and post-LayerNorm activations / GELU outputs / N(0, 0.02)-scale
        # weights sit well inside fp16's range.  The LayerNorms, the attention kernel, the fusion op and the GELU epilogue write fp16 twins
        # of their outputs, AdamW keeps an fp16 shadow of the text weights, the dense outputs reach the LayerNorms in f32.  The backward
        # pass is unchanged (bf16 operands: gradients need the range).  Replaces round 3's split-precision text stream (2.5x the text
        # forward FLOPs for the same logit error).  MART_TEXT_F16=0: plain bf16 text stream (A/B).
        self.text_f16 = os.environ.get("MART_TEXT_F16", "1") == "1"
        # d(visual) of the fusion op through a side buffer (see backward): takes the fusion backward of text layers 8-10 off the main queue's
        # dependency chain.  Measured neutral (87.6 / 88.1 ms against 87.8 / 87.9 in place, same box: the step is throughput-bound, the queues only
        # fill each other's tails), so the in-place form stays the default; the switch and its test keep the schedule available.
        self.fusion_side = os.environ.get("MART_FUSION_SIDE", "0") == "1"
        self._w3cache: Dict[str, tuple] = {}
        # ln_fold (opt-in: MART_LN_FOLD=1 or engine.ln_fold = True): in forward passes that keep nothing for a backward pass (torch.no_grad(): validation /
        # test / serving in the bf16 configuration) the vision stream's LayerNorms are folded into the products that consume them --
        # LN(x) W^T + b = rstd (x (gamma o W)^T - mean s) + b'  (modeling_unimo.py:509 -> :223-225, :518 -> :284-286): the out-proj / fc2 epilogue that writes
        # the f32 residual stream also writes its bf16 copy and per-row partial sums, a 9.6 MB finalise launch makes mean / rstd, and the Q/K/V / fc1
        # product applies them in its epilogue; 23 of the 24 `ln_fwd_fast_k` passes (309 MB f32 in, 155 MB out each) disappear.  As accurate as the
        # unfused pass (tests/test_ln_fold_model_gpu.py) and +0.2-2.7 % (mean +1.3 %) on the bf16 evaluation pass (28.34 against 28.71 ms over five alternations: the producers'
        # extra 155 MB and the consumers' heavier epilogues eat most of the 1.75 ms) -- NOT the default, because with it a no_grad pass and the forward
        # pass of a training step stop being bit-identical (test_text_fp16_forward_vs_plain_bf16_text_stream) for that 1 %.  A training step cannot
        # use it at all: the weight-gradient GEMM needs the normalised activations as its operand (docs/LAB_r01-r05.md section 6).
        self.ln_fold = os.environ.get("MART_LN_FOLD", "0") == "1"
        self._foldcache: Dict[str, tuple] = {}
        # wgrad_lag: the backward pass of a vision layer does not wait for the weight-gradient stream before its LayerNorm-1 backward rewrites the bf16
        # gradient buffer the fc2 weight-gradient GEMM reads -- it writes a fresh buffer instead, and the weight-gradient queue may lag behind the main
        # queue by more than one layer (joined at the end of the pass).  Same kernels, same operands, bit-identical gradients; 85.2 -> 84.1 ms per step at
        # 196 patches (three alternations, profiles/r05_wgrad_lag_ab.txt), neutral at 49.  MART_WGRAD_LAG=0: the per-layer join of rounds 1-4.
        self.wgrad_lag = os.environ.get("MART_WGRAD_LAG", "1") == "1"
        # grad_stream_bf16 (round 6): the gradient w.r.t. the VISION residual stream is carried in bf16 between layers instead of f32 + a bf16 copy.  Every
        # consumer of that stream already read the bf16 copy (data- and weight-gradient GEMMs); the f32 tensor only fed the next LayerNorm backward's
        # residual add and the fusion op's in-place accumulation.  Those now read / update the bf16 tensor (f32 arithmetic, one rounding per half layer):
        # the vision LayerNorm backward moves 10 bytes per element instead of 16, the fusion backward's read-modify-write 4 instead of 10.  Measured on the
        # reference golden G7 (emulation on the f32 path, tests/test_parity_full_gpu.py): worst gradient-norm deviation unchanged (7.8e-3 over 431 tensors),
        # worst sampled rel-L2 1.33e-2 -> 1.49e-2 -- the bf16 GEMM operands dominate.  The forward residual streams stay f32.  MART_GRAD_STREAM_BF16=0: f32.
        self.grad_stream_bf16 = os.environ.get("MART_GRAD_STREAM_BF16", "1") == "1"

    # ------------------------------------------------------------------ helpers
    def _lin(self, name):
        st = self.st
        return st.w(name + ".weight"), st.m(name + ".bias")

    def _w3(self, *names: str) -> torch.Tensor:
        """[sum(out), 3*in] two-term bf16 split [hi|hi|lo] of one (or several adjacent) f32 master weights, cached until the weights change."""
        ver = self.st.version
        hit = self._w3cache.get(names[0])
        if hit is None or hit[0] != ver:
            W = self.st.fused(list(names), self.st.master)
            hit = (ver, ops.split_bf16x3(W.view(W.shape[0], -1), 1))
            self._w3cache[names[0]] = hit
        return hit[1]

    def _folded(self, wnames, bnames, ln: str):
        """(gamma o W as bf16, s[n] = its row sums, b' = b + W beta) of the linear layer(s) ``wnames`` behind LayerNorm ``ln``; cached until the weights change."""
        st = self.st
        hit = self._foldcache.get(wnames[0])
        if hit is None or hit[0] != st.version:
            W = st.fused(list(wnames), st.master)
            b = st.fused(list(bnames), st.master)
            Wf = torch.empty(W.shape, device=W.device, dtype=BF)
            s_, bf_ = torch.empty(W.shape[0], device=W.device, dtype=F32), torch.empty(W.shape[0], device=W.device, dtype=F32)
            ops.ln_fold_prep(W, b, st.m(ln + ".weight"), st.m(ln + ".bias"), Wf, s_, bf_)
            hit = (st.version, Wf, s_, bf_)
            self._foldcache[wnames[0]] = hit
        return hit[1], hit[2], hit[3]

    def _wgrad(self, X, Y, wname, bname=None, NX=None):
        """dW[wname] += X^T Y ; db[bname] += colsum(X)."""
        g = self.st.g(wname)
        self._tn(X, Y, g.view(g.shape[0], -1), NX=NX, colsum=self.st.g(bname) if bname else None)

    def _tn(self, X, Y, out, **kw):
        """Weight-gradient GEMM.  With ``overlap_wgrad`` it is issued on a side stream: it only feeds the flat gradient
        buffer, so it can run next to the data-gradient chain; the two kernels' HBM-bound epilogues and MFMA-bound main
        loops then interleave across the CUs instead of alternating in lockstep."""
        if not self.overlap_wgrad:
            ops.gemm_tn(X, Y, out, **kw)
            return
        if self._side is None:
            self._side = torch.cuda.Stream(priority=int(os.environ.get("MART_WGRAD_PRIO", "0")))
        main = torch.cuda.current_stream()
        side = self._side
        if self._wg2 and self._tstream is not None and main == self._tstream:
            # weight gradients issued from the text stream get their own queue: in ONE in-order queue a vision weight-gradient GEMM whose operands
            # are ready waits behind a text one whose event the text stream has not reached yet
            if self._side_t is None:
                self._side_t = torch.cuda.Stream(priority=int(os.environ.get("MART_WGRAD_PRIO", "0")))
            side = self._side_t
            self._side_t_busy = True
        else:
            self._side_busy = True
        ev = torch.cuda.Event()
        ev.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            ops.gemm_tn(X, Y, out, **kw)
        X.record_stream(side)
        Y.record_stream(side)

    def _ln_bwd(self, **kw):
        """LayerNorm backward.  Its ordered dgamma / dbeta reduction (a 14 us launch that only feeds the flat gradient buffer) is issued on the
        weight-gradient stream instead of sitting between two kernels of the data-gradient chain with ``MART_LN_DEFER=1``.  Measured on the bench
        step (same box, interleaved, twice): 88.53 ms deferred against 88.08 in order -- the reduction is cheaper inside the chain than the event +
        cross-stream wait it needs outside; off by default, kept for the A/B and covered by tests."""
        if not (self.overlap_wgrad and self.ln_defer and ops.TN_DETERMINISTIC and (kw.get("dgamma") is not None or kw.get("dbeta") is not None)):
            ops.ln_bwd(**kw)
            return
        ws, n = ops.ln_bwd(defer_reduce=True, **kw)
        if self._side is None:
            self._side = torch.cuda.Stream(priority=int(os.environ.get("MART_WGRAD_PRIO", "0")))
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        with torch.cuda.stream(self._side):
            self._side.wait_event(ev)
            ops.ln_dgb_reduce(ws, n, kw["H"], kw.get("dgamma"), kw.get("dbeta"))
        ws.record_stream(self._side)
        self._side_busy = True

    # ---- text stream (two_stream): the text layers are issued on their own HIP stream; events mirror the cross-wiring of
    # the model (text K/V -> vision attention, vision output -> text fusion, and the reverse edges in backward), so small
    # text kernels fill the gaps of the big vision kernels instead of running alone on the GPU.  With two_stream off every
    # helper is a no-op and both "streams" are the current one: one code path.
    def _text_ctx(self):
        import contextlib
        return torch.cuda.stream(self._tstream) if self._tstream is not None else contextlib.nullcontext()

    def _text_begin(self):
        if not self.two_stream:
            self._tstream = None
        elif self._tstream is None:
            self._tstream = torch.cuda.Stream(priority=int(os.environ.get("MART_TEXT_PRIO", "0")))
        if self._tstream is not None:
            self._tstream.wait_stream(torch.cuda.current_stream())

    def _text_done(self):
        if self._tstream is not None:
            torch.cuda.current_stream().wait_stream(self._tstream)

    def _text_record(self):
        if self._tstream is None:
            return None
        ev = torch.cuda.Event()
        ev.record(self._tstream)
        return ev

    def _main_record(self):
        if self._tstream is None:
            return None
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        return ev

    def _text_wait(self, ev):
        if self._tstream is not None and ev is not None:
            self._tstream.wait_event(ev)

    def _main_wait(self, ev):
        if self._tstream is not None and ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def _text_uses(self, t):
        """A tensor allocated on the main stream that text-stream kernels will touch (allocator must not recycle it early)."""
        if self._tstream is not None:
            t.record_stream(self._tstream)

    def _join(self):
        """Main stream waits for every weight-gradient GEMM issued so far."""
        if self._side_busy:
            torch.cuda.current_stream().wait_stream(self._side)
            self._side_busy = False
        if self._side_t_busy:
            torch.cuda.current_stream().wait_stream(self._side_t)
            self._side_t_busy = False

    def _mark(self, name: str) -> None:
        """Timeline marks (tools/tail_marks.py): a HIP event on the current stream when ``self.marks`` is a list; free otherwise."""
        m = getattr(self, "marks", None)
        if m is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream())
            import time
            m.append((name, ev, time.perf_counter()))

    def _pass_begin(self):
        while self.max_inflight > 0 and len(self._inflight) >= self.max_inflight:
            self._inflight.popleft().synchronize()

    def _pass_end(self):
        self._wg2 = False                           # (set per backward pass; a weight-gradient launch outside one uses the first side queue)
        if self.max_inflight > 0:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._inflight.append(ev)

    # ------------------------------------------------------------------ forward
    def forward(self, input_ids, attention_mask, token_type_ids, pixel_values, sep_idx, train: bool, seed: int,
                image_table=None, image_index=None, rows=None):
        """``rows`` (int32 [B, nr], flat row ids b * L + position): the caller promises to read only these rows of
        ``trans_hidden_states`` (lit_models/transformer.py:94-95,103-107: the [MASK] row and four more per example).  Nothing else reads
        the last text layer's other rows (modeling_unimo.py:616 exports the K/V of layers idx - 1 <= 10 only), so everything behind its
        attention / fusion -- output projection, FFN, both LayerNorms, the head transform, and their backward -- runs on B * nr rows
        instead of B * L.  Exact: the computed rows are what the dense pass computes.  With ``rows`` the returned tensors are COMPACT:
        ``trans`` f32 [B * nr, H] and ``trans_bf16`` [B * nr, H], slot order = ``rows`` (the dense [B, L, H] view with NaN elsewhere is built by the
        caller, functional._DenseRowsFn); ``backward`` then takes the gradient in the same compact layout."""
        self._pass_begin()
        self._mark("fwd_begin")
        st, H, nh, I = self.st, self.H, self.nh, self.I
        dev = input_ids.device
        B, Lq = input_ids.shape
        S, p = self.vc.image_size, self.vc.patch_size
        P = (S // p) ** 2
        Nv = 1 + 2 * P
        Mv, Mt = B * Nv, B * Lq
        Nvp = ((Nv + 63) // 64) * 64
        sv: Dict[str, object] = dict(B=B, L=Lq, P=P, Nv=Nv, Nvp=Nvp, train=train, seed=seed,
                                     ids=input_ids, tt=token_type_ids, am=attention_mask, sep=sep_idx)
        p_h = self.p_hidden if train else 0.0
        keep = bool(getattr(self, "save_for_backward", True))      # False under torch.no_grad(): backward-only outputs are skipped
        p_a = self.p_attn if train else 0.0
        f16 = self.text_f16
        R = nr = None
        if self.inject is not None or self.inject_grad is not None:
            rows = None                               # teacher forcing replaces whole [B, L, H] streams: dense pass
        if rows is not None:
            assert rows.dtype == torch.int32 and rows.dim() == 2 and rows.shape[0] == B and rows.is_contiguous()
            R, nr = rows.view(-1), int(rows.shape[1])
            sv["rows"] = (R, nr)
        Mr = B * nr if nr else 0
        self._text_begin()                            # text stream starts behind whatever produced the inputs

        # ---- vision embeddings: patchify -> GEMM -> assemble(+cls,+pos) -> pre-LN    (modeling_unimo.py:119-132,711)
        Kp = 3 * p * p
        patches = _e((B * 2 * P, Kp), BF, dev)
        if image_index is not None:                   # device-side batch assembly: gather straight from the resident image table
            assert image_table is not None and image_table.dtype == F32 and tuple(image_table.shape[1:]) == (3, S, S)
            assert image_index.dtype == torch.int32 and tuple(image_index.shape) == (B, 2)
            ops.patchify_gather(image_table, image_index.contiguous(), patches, B, S, p)
        else:
            pix = pixel_values.contiguous()
            assert pix.dtype == F32 and tuple(pix.shape) == (B, 2, 3, S, S), "pixel_values must be f32 [B,2,3,S,S]"
            ops.patchify(pix, patches, B, S, p)
        pe = _e((B * 2 * P, H), BF, dev)
        ops.gemm_nt(patches, st.w("unimo.vision_embeddings.patch_embedding.weight").view(H, Kp), pe)
        s_v = _e((Mv, H), F32, dev)
        ops.vision_assemble(pe, st.m("unimo.vision_embeddings.class_embedding"), st.m("unimo.vision_embeddings.position_embedding.weight"),
                            s_v, B, P, H)
        xv = _e((Mv, H), F32, dev)
        mean, rstd = _e((Mv,), F32, dev), _e((Mv,), F32, dev)
        ops.ln_fwd(x_f32=s_v, gamma=st.m("unimo.vision_pre_layrnorm.weight"), beta=st.m("unimo.vision_pre_layrnorm.bias"), eps=self.eps_v,
                   M=Mv, H=H, mean=mean, rstd=rstd, out_f32=xv)
        sv["vemb"] = (patches, s_v, mean, rstd)

        # ---- text embeddings (modeling_unimo.py:152-186)
        with self._text_ctx():
            u = "unimo.text_embeddings."
            s_t, tmean, trstd = _e((Mt, H), F32, dev), _e((Mt,), F32, dev), _e((Mt,), F32, dev)
            xt, xtb = _e((Mt, H), F32, dev), (_e((Mt, H), BF, dev) if keep or not f16 else None)
            xth = _e((Mt, H), HF, dev) if f16 else None
            ops.text_embed_fwd(ids=input_ids, tt=token_type_ids, word=st.m(u + "word_embeddings.weight"), pos=st.m(u + "position_embeddings.weight"),
                               type_=st.m(u + "token_type_embeddings.weight"), gamma=st.m(u + "LayerNorm.weight"), beta=st.m(u + "LayerNorm.bias"),
                               eps=self.eps_t, p_drop=p_h, seed=seed + 1, B=B, Lq=Lq, H=H, s_out=s_t, mean=tmean, rstd=trstd, out_f32=xt, out_bf16=xtb, out_f16=xth)
            sv["temb"] = (s_t, tmean, trstd)
        def tap_text(name, t):
            """Snapshot of the text stream, taken ON the text queue (that is where ``t`` is produced); the caller reads it on the main queue after the pass."""
            with self._text_ctx():
                c = t.view(B, Lq, H).clone()
            if self._tstream is not None:
                c.record_stream(torch.cuda.current_stream())
            self.taps[name] = c

        if self.taps is not None:
            self.taps["vis_emb"] = xv.view(B, Nv, H).clone()
            tap_text("txt_emb", xt)
        if self.inject is not None:
            assert not self.two_stream and not self.overlap_wgrad, "teacher forcing runs on the single-stream schedule"
            if "vis_emb" in self.inject:
                xv = self.inject["vis_emb"].reshape(Mv, H).to(device=dev, dtype=F32).contiguous()
            if "txt_emb" in self.inject:
                xt = self.inject["txt_emb"].reshape(Mt, H).to(device=dev, dtype=F32).contiguous()
                xtb, xth = xt.to(BF), (xt.to(HF) if f16 else None)

        t_qkv_prev = None
        ev_tqkv = ev_vis = None
        # (the fold's epilogues exist on gemm_nt's aligned fast path only: 16-byte aligned rows -- H a multiple of 128 covers every operand here -- and operands
        # below 2^31 elements; anything else takes the separate LayerNorm pass instead of an error from the kernel library)
        fold = self.ln_fold and not keep and self.taps is None and self.inject is None and H % 128 == 0 and Mv * max(I, 3 * H) < 2 ** 31
        xv_b = xv_stats = None                                         # bf16 copy / per-row partial sums of xv, when the previous layer's fc2 epilogue wrote them
        for l in range(self.n_layers):
            # ================= vision layer l (CLIPEncoderLayer.forward, modeling_unimo.py:490-527)
            v = f"unimo.encoder.vision_layers.{l}."
            m1, r1 = _e((Mv,), F32, dev), _e((Mv,), F32, dev)
            qkv = _e((Mv, 3 * H), BF, dev)
            names = [v + f"self_attn.{n}" for n in ("q_proj", "k_proj", "v_proj")]
            if fold and xv_stats is not None:                          # layer_norm1 folded into Q/K/V: xv_b / xv_stats come from the fc2 epilogue of layer l - 1
                h1 = None
                ops.ln_stats_finalize(xv_stats, Mv, H, self.eps_v, m1, r1)
                Wf, cs, bfold = self._folded([n + ".weight" for n in names], [n + ".bias" for n in names], v + "layer_norm1")
                ops.gemm_nt(xv_b, Wf, qkv, bias=bfold, ln_mean=m1, ln_rstd=r1, ln_colsum=cs)
            else:
                h1 = _e((Mv, H), BF, dev)
                ops.ln_fwd(x_f32=xv, gamma=st.m(v + "layer_norm1.weight"), beta=st.m(v + "layer_norm1.bias"), eps=self.eps_v, M=Mv, H=H,
                           mean=m1, rstd=r1, out_bf16=h1)
                ops.gemm_nt(h1, st.fused([n + ".weight" for n in names]), qkv, bias=st.fused([n + ".bias" for n in names], st.master))
            ctx, lse = _e((Mv, H), BF, dev), _e((B, nh, Nv), F32, dev)
            pre = t_qkv_prev if l >= self.fuse_from else None
            if pre is not None:
                self._main_wait(ev_tqkv)                                   # K/V of text layer l-1 are produced on the text stream
            akw = dict(q=qkv[:, :H], k=qkv[:, H:2 * H], v=qkv[:, 2 * H:], ctx=ctx, lse=lse, B=B, nh=nh, Sq=Nv, Sk=Nv, scale=0.125,
                       pk=pre[:, H:2 * H] if pre is not None else None, pv=pre[:, 2 * H:] if pre is not None else None,
                       Lp=Lq if pre is not None else 0)
            ops.attn_fwd(**akw)
            x1 = _e((Mv, H), F32, dev)
            w, b = self._lin(v + "self_attn.out_proj")
            m2, r2 = _e((Mv,), F32, dev), _e((Mv,), F32, dev)
            # z holds act'(fc1 output) (the only thing the backward pass needs of it); nothing is written under no_grad
            z, f = (_e((Mv, I), BF, dev) if keep else None), _e((Mv, I), BF, dev)
            if fold:                                                   # layer_norm2 folded into fc1
                h2 = None
                x1b, part = _e((Mv, H), BF, dev), _e((Mv, H // 64, 2), F32, dev)
                ops.gemm_nt(ctx, w, x1, bias=b, res_f32=xv, C2=x1b, row_stats=part)
                ops.ln_stats_finalize(part, Mv, H, self.eps_v, m2, r2)
                Wf, cs, bfold = self._folded([v + "mlp.fc1.weight"], [v + "mlp.fc1.bias"], v + "layer_norm2")
                ops.gemm_nt(x1b, Wf, f, bias=bfold, act=ops.ACT_QGELU, ln_mean=m2, ln_rstd=r2, ln_colsum=cs)
                del x1b, part
            else:
                ops.gemm_nt(ctx, w, x1, bias=b, res_f32=xv)
                h2 = _e((Mv, H), BF, dev)
                ops.ln_fwd(x_f32=x1, gamma=st.m(v + "layer_norm2.weight"), beta=st.m(v + "layer_norm2.bias"), eps=self.eps_v, M=Mv, H=H,
                           mean=m2, rstd=r2, out_bf16=h2)
                w, b = self._lin(v + "mlp.fc1")
                ops.gemm_nt(h2, w, f, bias=b, act=ops.ACT_QGELU, preact=z, preact_grad=keep)
            x2 = _e((Mv, H), F32, dev)
            nxt = fold and l + 1 < self.n_layers                      # the next layer's layer_norm1 is folded: it reads the bf16 copy and the partial sums
            x2b = _e((Mv, H), BF, dev) if (l >= self.fuse_from or nxt) else None
            xv_stats = _e((Mv, H // 64, 2), F32, dev) if nxt else None
            w, b = self._lin(v + "mlp.fc2")
            ops.gemm_nt(f, w, x2, bias=b, res_f32=x1, C2=x2b, row_stats=xv_stats)
            xv_b = x2b
            sv[f"v{l}"] = dict(x=xv, m1=m1, r1=r1, h1=h1, qkv=qkv, ctx=ctx, lse=lse, x1=x1, m2=m2, r2=r2, h2=h2, z=z, f=f, pre=pre)
            xv = x2

            if l >= self.fuse_from:
                ev_vis = self._main_record()                               # x2b (input of the fusion of text layer l) is final
            # ================= text layer l (BertLayer.forward, modeling_unimo.py:540-577)
            with self._text_ctx():
                t = f"unimo.encoder.text_layer.{l}."
                sub = R is not None and l == self.n_layers - 1           # last layer: post-attention part on the requested rows only
                tqkv = _e((Mt, 3 * H), BF, dev)
                names = [t + f"attention.self.{n}" for n in ("query", "key", "value")]
                qbias = st.fused([n + ".bias" for n in names], st.master)
                if f16:
                    ops.gemm_nt(xth, st.h(*[n + ".weight" for n in names]), tqkv, bias=qbias)
                else:
                    ops.gemm_nt(xtb, st.fused([n + ".weight" for n in names]), tqkv, bias=qbias)
                if l >= self.export_from:
                    ev_tqkv = self._text_record()
                tctx, tlse = _e((Mt, H), BF, dev), _e((B, nh, Lq), F32, dev)
                tctx_h = _e((Mt, H), HF, dev) if f16 else None
                w0 = st.m(t + "attention.self.adaptive_weight.0")
                w1 = st.m(t + "attention.self.adaptive_weight.1")
                tkw = dict(q=tqkv[:, :H], k=tqkv[:, H:2 * H], v=tqkv[:, 2 * H:], ctx=tctx, lse=tlse, B=B, nh=nh, Sq=Lq, Sk=Lq, scale=0.125,
                           attn_mask=attention_mask, sep=sep_idx[:, 2:] if sep_idx is not None else None,
                           sep_stride=sep_idx.shape[1] if sep_idx is not None else 0,
                           w0=w0 if sep_idx is not None else None, w1=w1 if sep_idx is not None else None,
                           p_drop=p_a, seed=seed + 10 + 4 * l)
                ops.attn_fwd(ctx_f16=tctx_h, **tkw)
                fus = fus_h = probs = visT = None
                if l >= self.fuse_from:                                   # BertFusion.forward, modeling_unimo.py:400-414
                    self._text_wait(ev_vis)
                    probs, visT = _e((Mt, Nvp), BF, dev), None
                    fus, fus_h = _e((Mt, H), BF, dev), (_e((Mt, H), HF, dev) if f16 else None)
                    if self.fused_fusion and ops.fusion_supported(Lq, Nv, H):
                        ops.fusion_fwd(tctx, x2b, fus, probs, B, Lq, Nv, H, out_f16=fus_h)   # scores / softmax / probs @ visual in one kernel
                    else:
                        scores = _e((Mt, Nvp), F32, dev)
                        ops.gemm_nt(tctx, x2b, scores, M=Lq, N=Nv, batch=B, stride_a=Lq * H, stride_b=Nv * H, stride_c=Lq * Nvp)
                        ops.softmax_fwd(scores, probs, Mt, Nv)
                        visT = _e((B * H, Nvp), BF, dev)
                        ops.transpose_bf16(x2b, visT, Nv, H, Nvp, batch=B, stride_i=Nv * H, stride_o=H * Nvp)
                        ops.gemm_nt(probs, visT, fus, M=Lq, N=H, batch=B, stride_a=Lq * Nvp, stride_b=H * Nvp, stride_c=Lq * H)
                        if f16:
                            ops.cast_bf16_f16(fus, fus_h)
                # ---- from here on: Ms rows (all B * L, or the B * nr requested ones of the last layer, gathered by the GEMM / LayerNorm reads)
                Ms, xr = (Mr, R) if sub else (Mt, None)
                w, b = self._lin(t + "attention.output.dense")
                s1, am1, ar1 = _e((Ms, H), F32, dev), _e((Ms,), F32, dev), _e((Ms,), F32, dev)
                a = _e((Ms, H), F32, dev)
                ab = _e((Ms, H), BF, dev) if keep or not f16 else None     # bf16 copies: what the backward pass multiplies
                ah = _e((Ms, H), HF, dev) if f16 else None
                lnkw = dict(gamma=st.m(t + "attention.output.LayerNorm.weight"), beta=st.m(t + "attention.output.LayerNorm.bias"), eps=self.eps_t,
                            M=Ms, H=H, mean=am1, rstd=ar1, s_out=s1, out_f32=a, out_bf16=ab, out_f16=ah, p_drop=p_h, seed=seed + 11 + 4 * l)
                if f16:                                                   # dense outputs reach the LayerNorm in f32
                    so = _e((Ms, H), F32, dev)
                    ops.gemm_nt(tctx_h, st.h(t + "attention.output.dense.weight"), so, bias=b, a_rows=xr)
                    ops.ln_fwd(x_f32=xt, x_rows=xr, y_f32=so, **lnkw)
                else:
                    so = _e((Ms, H), BF, dev)
                    ops.gemm_nt(tctx, w, so, bias=b, a_rows=xr)
                    ops.ln_fwd(x_f32=xt, x_rows=xr, y_bf16=so, **lnkw)
                fus_s, fus_sh = fus, fus_h
                if sub and fus is not None:                               # the fusion rows of the requested positions, compact
                    if keep or not f16:
                        fus_s = _e((Ms, H), BF, dev)
                        ops.gather_rows_bf16(fus, R, fus_s)
                    if f16:
                        fus_sh = _e((Ms, H), HF, dev)
                        ops.gather_rows_bf16(fus_h, R, fus_sh)            # (a 2-byte row copy: the element type does not matter)
                zt = _e((Ms, I), BF, dev) if keep else None
                w, b = self._lin(t + "intermediate.dense")
                bf_ = st.m(t + "intermediate.fusion_dense.bias") if fus is not None else None
                s2, om, orr = _e((Ms, H), F32, dev), _e((Ms,), F32, dev), _e((Ms,), F32, dev)
                xo = _e((Ms, H), F32, dev)
                xob = _e((Ms, H), BF, dev) if keep or not f16 or not self.head_split else None
                xoh = _e((Ms, H), HF, dev) if f16 and l < self.n_layers - 1 else None    # (the head transform reads the f32 stream)
                lnkw = dict(gamma=st.m(t + "output.LayerNorm.weight"), beta=st.m(t + "output.LayerNorm.bias"), eps=self.eps_t,
                            M=Ms, H=H, mean=om, rstd=orr, s_out=s2, out_f32=xo, out_bf16=xob, out_f16=xoh, p_drop=p_h, seed=seed + 12 + 4 * l)
                if f16:
                    hth, ht = _e((Ms, I), HF, dev), (_e((Ms, I), BF, dev) if keep else None)   # GELU output: fp16 for the next product, bf16 for the backward pass
                    ops.gemm_nt(ah, st.h(t + "intermediate.dense.weight"), hth, A2=fus_sh,
                                B2=st.h(t + "intermediate.fusion_dense.weight") if fus is not None else None,
                                bias=b, bias2=bf_, act=ops.ACT_GELU, preact=zt, preact_grad=keep, C2=ht)
                    oo = _e((Ms, H), F32, dev)
                    ops.gemm_nt(hth, st.h(t + "output.dense.weight"), oo, bias=st.m(t + "output.dense.bias"))
                    ops.ln_fwd(x_f32=a, y_f32=oo, **lnkw)
                else:
                    ht = _e((Ms, I), BF, dev)
                    if fus is not None:
                        wf, _ = self._lin(t + "intermediate.fusion_dense")
                        ops.gemm_nt(ab, w, ht, A2=fus_s, B2=wf, bias=b, bias2=bf_, act=ops.ACT_GELU, preact=zt, preact_grad=keep)
                    else:
                        ops.gemm_nt(ab, w, ht, bias=b, act=ops.ACT_GELU, preact=zt, preact_grad=keep)
                    w, b = self._lin(t + "output.dense")
                    oo = _e((Ms, H), BF, dev)
                    ops.gemm_nt(ht, w, oo, bias=b)
                    ops.ln_fwd(x_f32=a, y_bf16=oo, **lnkw)
                sv[f"t{l}"] = dict(xb=xtb, qkv=tqkv, ctx=tctx, lse=tlse, tkw=tkw, fus=fus_s if fus is not None else None, probs=probs, visT=visT, visb=x2b,
                                   s1=s1, m1=am1, r1=ar1, ab=ab, zt=zt, ht=ht, s2=s2, m2=om, r2=orr, sub=sub)
                xt, xtb, xth = xo, xob, xoh
            t_qkv_prev = tqkv if l >= self.export_from else None
            if self.taps is not None:
                self.taps[f"vis{l}"] = xv.view(B, Nv, H).clone()
                if xt.shape[0] == Mt:
                    tap_text(f"txt{l}", xt)
            if self.inject is not None:
                if f"vis{l}" in self.inject:
                    xv = self.inject[f"vis{l}"].reshape(Mv, H).to(device=dev, dtype=F32).contiguous()
                if f"txt{l}" in self.inject:
                    xt = self.inject[f"txt{l}"].reshape(Mt, H).to(device=dev, dtype=F32).contiguous()
                    xtb, xth = xt.to(BF), (xt.to(HF) if f16 else None)

        self._mark("fwd_vision_end")
        # ---- MLM head transform (BertPredictionHeadTransform.forward, modeling_unimo.py:972-975)
        with self._text_ctx():
            hp = "cls.predictions.transform."
            Mh = xt.shape[0]                                # B * L, or the B * nr requested rows
            y, zh = _e((Mh, H), F32, dev), _e((Mh, H), BF, dev)
            w, b = self._lin(hp + "dense")
            if self.head_split:
                ops.gemm_nt(ops.split_bf16x3(xt, 0), self._w3(hp + "dense.weight"), y, bias=b, act=ops.ACT_GELU, preact=zh)
            else:
                ops.gemm_nt(xtb, w, y, bias=b, act=ops.ACT_GELU, preact=zh)
            trans, transb = _e((Mh, H), F32, dev), _e((Mh, H), BF, dev)
            hm, hr = _e((Mh,), F32, dev), _e((Mh,), F32, dev)
            ops.ln_fwd(x_f32=y, gamma=st.m(hp + "LayerNorm.weight"), beta=st.m(hp + "LayerNorm.bias"), eps=self.eps_t, M=Mh, H=H, mean=hm, rstd=hr,
                       out_f32=trans, out_bf16=transb)
            sv["head"] = (xtb, y, zh, hm, hr)
        self._text_done()
        self._mark("fwd_end")
        if self._tstream is not None:                  # allocated on the text stream, consumed by the caller on the main stream
            trans.record_stream(torch.cuda.current_stream())
            transb.record_stream(torch.cuda.current_stream())
        if not keep:
            self._pass_end()                           # forward-only pass (no_grad); otherwise the backward pass closes it
        return (trans.view(B, Lq, H) if Mh == Mt else trans), transb, sv

    # ------------------------------------------------------------------ backward
    def backward(self, sv, dtrans: torch.Tensor) -> None:
        """Accumulates d(loss)/d(param) into FlatStore.grad given d(loss)/d(trans_hidden_states) (f32 [B,L,H])."""
        st, H, nh, I = self.st, self.H, self.nh, self.I
        dev = dtrans.device
        # the side / text streams fork from this (main) stream: it has to have waited for an off-stream gradient zero-fill / W^T refresh
        # (FlatStore.pending, MART_ASYNC_STEP=1 with zero_grad() between forward and backward) before any of them writes a gradient
        st.join_pending()
        self._mark("bwd_begin")
        B, Lq, P, Nv, Nvp = sv["B"], sv["L"], sv["P"], sv["Nv"], sv["Nvp"]
        train, seed = sv["train"], sv["seed"]
        Mv, Mt = B * Nv, B * Lq
        p_h = self.p_hidden if train else 0.0
        self._wg2 = (4 * Mt >= Mv) if self.wgrad_two == "auto" else self.wgrad_two == "1"
        def notify(off):
            if self.grad_ready is not None:
                self.grad_ready(off)
            if self.grad_ready_async is not None:                      # the consumer waits for these instead of the main stream waiting
                evs = []
                if self._side_busy:
                    e = torch.cuda.Event(); e.record(self._side); evs.append(e)
                if self._side_t_busy:
                    e = torch.cuda.Event(); e.record(self._side_t); evs.append(e)
                if self._tstream is not None:
                    e = torch.cuda.Event(); e.record(self._tstream); evs.append(e)
                self.grad_ready_async(off, evs)

        # ---- head transform
        self._text_begin()                                            # text stream starts behind the main stream (dtrans is ready)
        dtrans = dtrans.contiguous()
        self._text_uses(dtrans)
        R, nr = sv.get("rows", (None, None))
        with self._text_ctx():
            xtb, y, zh, hm, hr = sv["head"]
            hp = "cls.predictions.transform."
            Mh = y.shape[0]                                               # B * L, or the B * nr rows the forward pass was asked for
            dtr = dtrans.view(-1, H)                                      # [B * L, H], or compact [B * nr, H] for a row-subset pass
            assert dtr.shape[0] == Mh, "backward: the gradient must have the layout forward() returned (compact for a row-subset pass)"
            dyb = _e((Mh, H), BF, dev)
            self._ln_bwd(dy_f32=dtr, s=y, mean=hm, rstd=hr, gamma=st.m(hp + "LayerNorm.weight"), M=Mh, H=H, ds_bf16=dyb,
                       dgamma=st.g(hp + "LayerNorm.weight"), dbeta=st.g(hp + "LayerNorm.bias"))
            dzh = _e((Mh, H), BF, dev)
            ops.act_bwd(dyb, zh, ops.ACT_GELU, dzh)
            self._wgrad(dzh, xtb, hp + "dense.weight", hp + "dense.bias")
            d_f32 = _e((Mh, H), F32, dev)                                 # gradient w.r.t. the text stream, f32 part
            ops.gemm_nt(dzh, st.wt("head"), d_f32)
            d_b16 = None                                                  # ... plus an optional bf16 part
        if self.grad_ready is not None:
            self._join(); self._text_done(); self._text_begin()
        notify(st.slots["unimo.encoder.text_layer.%d.attention.self.query.weight" % (self.n_layers - 1)].offset)

        # gradient w.r.t. the vision stream: bf16 only (grad_stream_bf16; needs the fused fusion kernels in every fusion layer and none of the debugging
        # hooks / the side-buffer schedule, which work on the f32 tensor), or f32 + its bf16 copy
        gb16 = (self.grad_stream_bf16 and self.taps is None and self.inject_grad is None and not self.fusion_side and
                all(sv[f"t{l}"]["fus"] is None or sv[f"t{l}"]["visT"] is None for l in range(self.n_layers)))
        if gb16:
            dxv, dxvb = None, torch.zeros((Mv, H), device=dev, dtype=BF)
        else:
            dxv = torch.zeros((Mv, H), device=dev, dtype=F32)
            dxvb = _e((Mv, H), BF, dev)
        ev_vdone = self._main_record()                                # dxv / dxvb exist
        ev_vattn = None
        # d(visual) of the fusion op of text layer l (l = 8 .. 10) through a SIDE buffer: fusion_bwd used to accumulate into dxv in place, i.e. it
        # could start only when vision layer l + 1 had finished its backward pass and vision layer l could start only when it had finished -- a
        # 0.25-0.3 ms text-sized kernel alone on the critical path, three times per step.  With the side buffer the text stream runs it while
        # vision layer l + 1 is still going, and that layer's LayerNorm-1 backward (the kernel that produces dxv) adds the buffer as a second
        # residual operand (+25 % bytes in one HBM-bound launch).  The top layer keeps the in-place form (its dxv is the zero buffer above).
        side_mode = (self.fusion_side and self._tstream is not None and self.inject_grad is None and self.taps is None and
                     self.fused_fusion and ops.fusion_supported(Lq, Nv, H))
        T = dict(d_f32=d_f32, d_b16=d_b16, ev_tfus={}, A={}, A_side={}, fresh=False, dxvb=dxvb)

        def will_side(l):
            t_ = sv.get(f"t{l}") if l >= 0 else None
            return bool(side_mode and t_ is not None and 0 <= l < self.n_layers - 1 and t_["fus"] is not None and t_["visT"] is None)

        def text_A(l):
            """Text layer l backward, first part: output LayerNorm, FFN, the fusion op (d(visual) -> dxv in place, or -> a side buffer)."""
            d_f32, d_b16 = T["d_f32"], T["d_b16"]
            if self.taps is not None and d_f32.shape[0] == Mt:
                self.taps[f"dtxt{l}"] = (d_f32 + d_b16.float() if d_b16 is not None else d_f32.clone()).view(B, Lq, H)
            if self.inject_grad is not None and f"txt{l}" in self.inject_grad:
                d_f32, d_b16 = self.inject_grad[f"txt{l}"].reshape(Mt, H).to(device=dev, dtype=F32).contiguous(), None
            with self._text_ctx():
                t = f"unimo.encoder.text_layer.{l}."
                s = sv[f"t{l}"]
                fused = s["fus"] is not None
                sub = bool(s["sub"])                                      # the post-attention part of this layer ran on the requested rows only
                Ms = d_f32.shape[0] if sub else Mt
                ds2, doo = _e((Ms, H), F32, dev), _e((Ms, H), BF, dev)
                self._ln_bwd(dy_f32=d_f32, dy_bf16=d_b16, s=s["s2"], mean=s["m2"], rstd=s["r2"], gamma=st.m(t + "output.LayerNorm.weight"), M=Ms, H=H,
                           ds_f32=ds2, ds_bf16=doo, p_drop=p_h, seed=seed + 12 + 4 * l,
                           dgamma=st.g(t + "output.LayerNorm.weight"), dbeta=st.g(t + "output.LayerNorm.bias"))
                self._wgrad(doo, s["ht"], t + "output.dense.weight", t + "output.dense.bias")
                dzt = _e((Ms, I), BF, dev)
                ops.gemm_nt(doo, st.wt(f"t{l}.out"), dzt, mulz=s["zt"], mul_act=ops.ACT_STORED)
                self._wgrad(dzt, s["ab"], t + "intermediate.dense.weight", t + "intermediate.dense.bias")
                da2 = _e((Ms, H), BF, dev)
                ops.gemm_nt(dzt, st.wt(f"t{l}.int"), da2)
                dctx_fus = side = None
                if fused:
                    self._wgrad(dzt, s["fus"], t + "intermediate.fusion_dense.weight", t + "intermediate.fusion_dense.bias")
                    if sub:                                                        # d(fusion_output) of the requested rows, zero elsewhere
                        dfr = _e((Ms, H), F32, dev)
                        ops.gemm_nt(dzt, st.wt(f"t{l}.fus"), dfr)
                        dfus = torch.zeros((Mt, H), device=dev, dtype=BF)
                        ops.scatter_rows(dfr, R, nr, dfus, accumulate=True)
                    else:
                        dfus = _e((Mt, H), BF, dev)
                        ops.gemm_nt(dzt, st.wt(f"t{l}.fus"), dfus)
                    if s["visT"] is None:                                          # fused kernel ran forward: its backward twin
                        dctx_fus = _e((Mt, H), BF, dev)
                        if will_side(l):
                            side = torch.zeros((Mv, H), device=dev, dtype=F32)     # consumed by vision layer l + 1's LayerNorm-1 backward
                            ops.fusion_bwd(s["ctx"], s["visb"], dfus, s["probs"], dctx_fus, side, None, B, Lq, Nv, H)
                        else:
                            self._text_wait(T["ev_vdone"])                         # dxv holds the gradient left by vision layer l+1
                            ops.fusion_bwd(s["ctx"], s["visb"], dfus, s["probs"], dctx_fus, dxv, T["dxvb"], B, Lq, Nv, H)   # d(vis) added in place (f32 stream: bf16 copy refreshed; bf16 stream: dxv is None)
                            T["fresh"] = True
                    else:
                        dprobs = _e((Mt, Nvp), F32, dev)
                        ops.gemm_nt(dfus, s["visb"], dprobs, M=Lq, N=Nv, batch=B, stride_a=Lq * H, stride_b=Nv * H, stride_c=Lq * Nvp)
                        dsc = _e((Mt, Nvp), BF, dev)
                        ops.softmax_bwd(s["probs"], dprobs, dsc, Mt, Nv)
                        dctx_fus = _e((Mt, H), BF, dev)
                        ops.gemm_nt(dsc, s["visT"], dctx_fus, M=Lq, N=H, batch=B, stride_a=Lq * Nvp, stride_b=H * Nvp, stride_c=Lq * H)
                        # d(vis) = dS^T ctx + P^T d(fus), accumulated into the vision-stream gradient
                        if Lq % 64 == 0:
                            # one batched NT product with two K segments and the fp32 accumulate fused as residual: the short
                            # contraction (Lq) makes the transposed-operand form ~2.5x cheaper than two split-1 TN launches with
                            # 65 k atomics per workgroup
                            dscT, prT = _e((B * Nv, Lq), BF, dev), _e((B * Nv, Lq), BF, dev)
                            ops.transpose_bf16(dsc, dscT, Lq, Nv, Lq, batch=B, stride_i=Lq * Nvp, stride_o=Nv * Lq)
                            ops.transpose_bf16(s["probs"], prT, Lq, Nv, Lq, batch=B, stride_i=Lq * Nvp, stride_o=Nv * Lq)
                            ctxT, dfT = _e((B * H, Lq), BF, dev), _e((B * H, Lq), BF, dev)
                            ops.transpose_bf16(s["ctx"], ctxT, Lq, H, Lq, batch=B, stride_i=Lq * H, stride_o=H * Lq)
                            ops.transpose_bf16(dfus, dfT, Lq, H, Lq, batch=B, stride_i=Lq * H, stride_o=H * Lq)
                            self._text_wait(T["ev_vdone"])                         # dxv holds the gradient left by vision layer l+1
                            ops.gemm_nt(dscT, ctxT, dxv, A2=prT, B2=dfT, M=Nv, N=H, batch=B, stride_a=Nv * Lq, stride_b=H * Lq,
                                        stride_c=Nv * H, stride_aux=Nv * H, res_f32=dxv, C2=dxvb)     # ... and refreshes the bf16 copy
                            T["fresh"] = True
                        else:
                            self._text_wait(T["ev_vdone"])
                            ops.gemm_tn(dsc, s["ctx"], dxv, M=Lq, NX=Nv, NY=H, batch=B, stride_x=Lq * Nvp, stride_y=Lq * H, stride_o=Nv * H, splits=1)
                            ops.gemm_tn(s["probs"], dfus, dxv, M=Lq, NX=Nv, NY=H, batch=B, stride_x=Lq * Nvp, stride_y=Lq * H, stride_o=Nv * H, splits=1)
                    T["ev_tfus"][l] = self._text_record()
            T["A"][l] = dict(ds2=ds2, da2=da2, dctx_fus=dctx_fus, Ms=Ms, sub=sub)
            T["A_side"][l] = side

        def text_B(l):
            """Second part: attention-output LayerNorm, attention, Q/K/V -> the gradient w.r.t. the layer's input."""
            a = T["A"].pop(l)
            ds2, da2, dctx_fus, Ms, sub = a["ds2"], a["da2"], a["dctx_fus"], a["Ms"], a["sub"]
            with self._text_ctx():
                t = f"unimo.encoder.text_layer.{l}."
                s = sv[f"t{l}"]
                ds1, dso = _e((Ms, H), F32, dev), _e((Ms, H), BF, dev)
                self._ln_bwd(dy_f32=ds2, dy_bf16=da2, s=s["s1"], mean=s["m1"], rstd=s["r1"], gamma=st.m(t + "attention.output.LayerNorm.weight"), M=Ms, H=H,
                           ds_f32=ds1, ds_bf16=dso, p_drop=p_h, seed=seed + 11 + 4 * l,
                           dgamma=st.g(t + "attention.output.LayerNorm.weight"), dbeta=st.g(t + "attention.output.LayerNorm.bias"))
                if sub:
                    ctx_r = _e((Ms, H), BF, dev)
                    ops.gather_rows_bf16(s["ctx"], R, ctx_r)
                    self._wgrad(dso, ctx_r, t + "attention.output.dense.weight", t + "attention.output.dense.bias")
                    dcr = _e((Ms, H), F32, dev)
                    ops.gemm_nt(dso, st.wt(f"t{l}.ao"), dcr)
                    dctx = dctx_fus if dctx_fus is not None else torch.zeros((Mt, H), device=dev, dtype=BF)
                    ops.scatter_rows(dcr, R, nr, dctx, accumulate=True)    # bf16(f32 product + bf16 fusion part): the dense path's one rounding
                else:
                    self._wgrad(dso, s["ctx"], t + "attention.output.dense.weight", t + "attention.output.dense.bias")
                    dctx = _e((Mt, H), BF, dev)
                    ops.gemm_nt(dso, st.wt(f"t{l}.ao"), dctx, res_bf16=dctx_fus)
                # attention backward; for layers whose (K,V) fed a vision layer the k/v blocks already hold the prefix grads
                has_prefix_grad = self.export_from <= l < self.n_layers - 1
                dqkv = s.get("dqkv")
                if dqkv is None:
                    dqkv = _e((Mt, 3 * H), BF, dev)
                delta = _e((B, nh, Lq), F32, dev)
                sep_on = s["tkw"]["sep"] is not None
                if has_prefix_grad:
                    self._text_wait(T["ev_vattn"])                         # prefix gradients written by vision layer l+1's attention backward
                ops.attn_bwd(dctx=dctx, delta=delta, dq=dqkv[:, :H], dk=dqkv[:, H:2 * H], dv=dqkv[:, 2 * H:], accum_dkv=has_prefix_grad,
                             dw=st.g(t + "attention.self.adaptive_weight.0") if sep_on else None, **s["tkw"])
                names = [t + f"attention.self.{n}" for n in ("query", "key", "value")]
                gw = st.fused([n + ".weight" for n in names], st.grad)
                self._tn(dqkv, s["xb"], gw, colsum=st.fused([n + ".bias" for n in names], st.grad))
                dtb = _e((Mt, H), BF, dev)
                ops.gemm_nt(dqkv, st.wt(f"t{l}.qkv"), dtb)
                if sub:                                                    # residual-path gradient of the requested rows into the [B * L, H] stream
                    d_f32 = torch.zeros((Mt, H), device=dev, dtype=F32)
                    ops.scatter_rows(ds1, R, nr, d_f32, accumulate=True)
                    T["d_f32"], T["d_b16"] = d_f32, dtb
                else:
                    T["d_f32"], T["d_b16"] = ds1, dtb

        def text_embed_bwd():
            """Text embeddings backward: dropout -> LN -> scatter into the embedding tables (text stream)."""
            with self._text_ctx():
                s_t, tmean, trstd = sv["temb"]
                u = "unimo.text_embeddings."
                dyd = _e((Mt, H), F32, dev)
                ops.dropout_bwd_f32(T["d_f32"], T["d_b16"], dyd, Mt * H, p_h, seed + 1)
                dse = _e((Mt, H), F32, dev)
                self._ln_bwd(dy_f32=dyd, s=s_t, mean=tmean, rstd=trstd, gamma=st.m(u + "LayerNorm.weight"), M=Mt, H=H, ds_f32=dse,
                             dgamma=st.g(u + "LayerNorm.weight"), dbeta=st.g(u + "LayerNorm.bias"))
                ops.text_embed_scatter(dse, sv["ids"], sv["tt"], st.g(u + "word_embeddings.weight"), st.g(u + "position_embeddings.weight"),
                                       st.g(u + "token_type_embeddings.weight"), B, Lq, H)

        T["ev_vdone"], T["ev_vattn"] = ev_vdone, ev_vattn
        text_A(self.n_layers - 1)
        for l in reversed(range(self.n_layers)):
            # ================= text layer l (its first part was issued ahead: right above for the top layer, inside vision layer l + 1 below)
            text_B(l)
            if l == 0:
                # the text stream ends here: embedding backward, then its tables (params.layout_order puts them in front of vision layer 0) are released
                # to the all-reduce / AdamW consumers while the vision stream still has layer 0 and its embeddings to go
                text_embed_bwd()
                if self.grad_ready is not None:
                    self._join(); self._text_done(); self._text_begin()
                notify(st.slots["unimo.encoder.vision_layers.0.self_attn.q_proj.weight"].offset)
            side_next = None                                           # d(visual) of the fusion of text layer l - 1, when it travels through a side buffer
            # ================= vision layer l
            if l in T["ev_tfus"] and T["A_side"].get(l) is None:
                self._main_wait(T["ev_tfus"][l])                           # text layer l added d(vis) into dxv in place
            dxvb_fresh = T["fresh"] or T["A_side"].get(l) is not None      # in place: fusion_bwd refreshed the bf16 copy; side buffer: LayerNorm-1 of layer l + 1 did
            T["fresh"] = False
            v = f"unimo.encoder.vision_layers.{l}."
            s = sv[f"v{l}"]
            self._mark(f"bwd_v{l}")
            if self.taps is not None:
                self.taps[f"dvis{l}"] = dxv.view(B, Nv, H).clone()
            if self.inject_grad is not None and f"vis{l}" in self.inject_grad:
                dxv.copy_(self.inject_grad[f"vis{l}"].reshape(Mv, H))
                dxvb_fresh = False
                ops.add_f32_bf16(dxv, None, None, dxvb)
            elif not gb16 and (l >= self.fuse_from or l == self.n_layers - 1) and not dxvb_fresh:   # fusion of text layer l added d(vis) into dxv
                ops.add_f32_bf16(dxv, None, None, dxvb)                # -> refresh the bf16 copy (otherwise ln1 bwd wrote it)
            self._wgrad(dxvb, s["f"], v + "mlp.fc2.weight", v + "mlp.fc2.bias")
            dz = _e((Mv, I), BF, dev)
            ops.gemm_nt(dxvb, st.wt(f"v{l}.fc2"), dz, mulz=s["z"], mul_act=ops.ACT_STORED)
            self._wgrad(dz, s["h2"], v + "mlp.fc1.weight", v + "mlp.fc1.bias")
            dh2 = _e((Mv, H), BF, dev)
            ops.gemm_nt(dz, st.wt(f"v{l}.fc1"), dh2)
            del dz
            dx1, dx1b = (None if gb16 else _e((Mv, H), F32, dev)), _e((Mv, H), BF, dev)
            self._ln_bwd(dy_bf16=dh2, s=s["x1"], mean=s["m2"], rstd=s["r2"], gamma=st.m(v + "layer_norm2.weight"), M=Mv, H=H,
                       add_f32=dxv, add_bf16=dxvb if gb16 else None,
                       ds_f32=dx1, ds_bf16=dx1b, bf16_total=True, dgamma=st.g(v + "layer_norm2.weight"), dbeta=st.g(v + "layer_norm2.bias"))
            self._wgrad(dx1b, s["ctx"], v + "self_attn.out_proj.weight", v + "self_attn.out_proj.bias")
            dctx = dh2                                                 # reuse
            ops.gemm_nt(dx1b, st.wt(f"v{l}.o"), dctx)
            dqkv = _e((Mv, 3 * H), BF, dev)
            delta = _e((B, nh, Nv), F32, dev)
            pre = s["pre"]
            dpre = None
            if pre is not None:                                        # prefix grads land in text layer l-1's dqkv buffer (k,v blocks)
                dpre = _e((Mt, 3 * H), BF, dev)
                sv[f"t{l - 1}"]["dqkv"] = dpre
                self._text_uses(dpre)
            q = s["qkv"]
            ops.attn_bwd(dctx=dctx, delta=delta, dq=dqkv[:, :H], dk=dqkv[:, H:2 * H], dv=dqkv[:, 2 * H:],
                         dpk=dpre[:, H:2 * H] if dpre is not None else None, dpv=dpre[:, 2 * H:] if dpre is not None else None,
                         q=q[:, :H], k=q[:, H:2 * H], v=q[:, 2 * H:], ctx=s["ctx"], lse=s["lse"], B=B, nh=nh, Sq=Nv, Sk=Nv, scale=0.125,
                         pk=pre[:, H:2 * H] if pre is not None else None, pv=pre[:, 2 * H:] if pre is not None else None,
                         Lp=Lq if pre is not None else 0)
            T["ev_vattn"] = self._main_record()
            names = [v + f"self_attn.{n}" for n in ("q_proj", "k_proj", "v_proj")]
            self._tn(dqkv, s["h1"], st.fused([n + ".weight" for n in names], st.grad), colsum=st.fused([n + ".bias" for n in names], st.grad))
            dh1 = dctx
            ops.gemm_nt(dqkv, st.wt(f"v{l}.qkv"), dh1)
            del dqkv
            early = will_side(l - 1)
            if early:
                # first part of text layer l - 1 (through its fusion op), issued here so that its side-buffer d(visual) can enter the LayerNorm-1
                # backward below; it depends on text layer l's second part (issued above) and on nothing of this vision layer
                text_A(l - 1)
                side_next = T["A_side"][l - 1]
                self._main_wait(T["ev_tfus"][l - 1])
                side_next.record_stream(torch.cuda.current_stream())   # allocated on the text stream, read here
            if self.wgrad_lag and self.overlap_wgrad:
                # a fresh bf16 buffer for the gradient this LayerNorm writes instead of a join: the fc2 weight-gradient GEMM of this layer may still
                # be reading the old one on the weight-gradient stream (the caching allocator keeps it until that stream has passed it: record_stream
                # in _tn), and the main queue goes on while the weight-gradient queue lags behind by more than one layer
                dxvb = _e((Mv, H), BF, dev)
            else:
                self._join()                                           # dxvb (read by the fc2 weight-gradient GEMM) is rewritten next
            self._ln_bwd(dy_bf16=dh1, s=s["x"], mean=s["m1"], rstd=s["r1"], gamma=st.m(v + "layer_norm1.weight"), M=Mv, H=H, add_f32=dx1,
                       add_bf16=dx1b if gb16 else None, add2_f32=side_next, ds_f32=dxv, ds_bf16=dxvb, bf16_total=True,
                       dgamma=st.g(v + "layer_norm1.weight"), dbeta=st.g(v + "layer_norm1.bias"))
            T["dxvb"] = dxvb                                           # (a fresh buffer under wgrad_lag: what text layer l - 1's fusion backward updates)
            if early:
                T["A_side"][l - 1] = True                              # consumed: keep the "a side buffer existed" mark, drop the [Mv, H] f32 tensor
                del side_next
            T["ev_vdone"] = self._main_record()
            if l > 0 and not early:
                text_A(l - 1)                                          # (in-place form: its fusion backward adds into the dxv written just above)
            sv[f"v{l}"] = None
            sv[f"t{l}"] = None
            if l > 0:
                if self.grad_ready is not None:
                    self._join(); self._text_done(); self._text_begin()
                notify(st.slots[f"unimo.encoder.text_layer.{l - 1}.attention.self.query.weight"].offset)

        # ---- vision embeddings backward: pre-LN -> assemble -> patch GEMM weight gradient
        patches, s_v, vmean, vrstd = sv["vemb"]
        dsv = _e((Mv, H), F32, dev)
        self._ln_bwd(dy_f32=dxv, dy_bf16=dxvb if gb16 else None, s=s_v, mean=vmean, rstd=vrstd, gamma=st.m("unimo.vision_pre_layrnorm.weight"), M=Mv, H=H, ds_f32=dsv,
                   dgamma=st.g("unimo.vision_pre_layrnorm.weight"), dbeta=st.g("unimo.vision_pre_layrnorm.bias"))
        dpe = _e((B * 2 * P, H), BF, dev)
        ops.vision_assemble_bwd(dsv, dpe, st.g("unimo.vision_embeddings.class_embedding"), st.g("unimo.vision_embeddings.position_embedding.weight"),
                                B, P, H)
        gw = st.g("unimo.vision_embeddings.patch_embedding.weight")
        self._tn(dpe, patches, gw.view(H, -1))
        self._mark("bwd_embed_end")
        self._text_done()                                             # main stream waits for the text stream
        self._join()
        self._mark("bwd_end")
        notify(st.total)
        self._pass_end()
