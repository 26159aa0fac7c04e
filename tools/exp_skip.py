"""Upper bounds by knock-out: run bench.py's training loop with selected kernels skipped (results are garbage on purpose).
usage: python tools/exp_skip.py [text_attn] [text_ln] [fusion] ...   -- prints examples/s and ms/step"""
import os, sys, json, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mkg_analogy_amd import ops
what = set(sys.argv[1:])
sys.argv = [sys.argv[0], "--steps", "15", "--warmup", "4", "--no-cpu-baseline", "--no-kernel-timing", "--train-only"]
o_fwd, o_bwd, o_lnf, o_lnb, o_ff, o_fb = ops.attn_fwd, ops.attn_bwd, ops.ln_fwd, ops.ln_bwd, ops.fusion_fwd, ops.fusion_bwd
if "text_attn" in what:
    ops.attn_fwd = lambda **kw: None if kw["Sq"] <= 128 else o_fwd(**kw)
    ops.attn_bwd = lambda **kw: None if kw["Sq"] <= 128 else o_bwd(**kw)
if "vis_attn_bwd" in what:
    ops.attn_bwd = lambda **kw: None if kw["Sq"] > 128 else o_bwd(**kw)
if "vis_attn_fwd" in what:
    ops.attn_fwd = lambda **kw: None if kw["Sq"] > 128 else o_fwd(**kw)
if "text_ln" in what:
    ops.ln_fwd = lambda **kw: None if kw["M"] < 50000 else o_lnf(**kw)
    ops.ln_bwd = lambda **kw: None if kw["M"] < 50000 else o_lnb(**kw)
if "vis_ln" in what:
    ops.ln_fwd = lambda **kw: None if kw["M"] >= 50000 else o_lnf(**kw)
    ops.ln_bwd = lambda **kw: None if kw["M"] >= 50000 else o_lnb(**kw)
if "fusion" in what:
    ops.fusion_fwd = lambda *a, **kw: None
    ops.fusion_bwd = lambda *a, **kw: None
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print(sorted(what), d["value"], d["ms_per_step"])
