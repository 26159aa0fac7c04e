"""Does any kernel of the forward pass read memory it did not write?  Run the same eval-mode forward twice, poisoning the caching allocator's
free blocks with NaN (and with large finite garbage) in between: outputs must be bit-identical.  usage: python tools/uninit_probe.py [g8b_pretrain_p196_cond|g7_bench_cond]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_parity_full_gpu as T

tag = sys.argv[1] if len(sys.argv) > 1 else "g8b_pretrain_p196_cond"
g = T._load(tag)
pre = bool(int(g["pretrain"]))
model, lit, cfg = T._product(g, pretrain=pre)
batch = T._batch(g, pretrain=pre)
gb = {k: v.cuda() for k, v in batch.items()}
keys = ("input_ids", "attention_mask", "token_type_ids", "pixel_values") + (() if pre else ("sep_idx",))
model.eval()


def fwd(plain=False, subset=False):
    """plain: MART_TEXT_F16=0 (bf16 text stream); subset: last text layer + head on the [MASK] rows only (needed_rows)."""
    model.engine.text_f16 = not plain
    with torch.no_grad():
        need = (gb["input_ids"] == 103).int().argmax(1) if subset else None
        out, trans = model(**{k: gb[k] for k in keys}, return_dict=True, needed_rows=need)
        rows = out.logits.mask_rows(gb["input_ids"], 103)
        r = rows[:, 30522:30522 + 11292].float().clone()
    model.engine.text_f16 = True
    return r, trans.float().clone()


def poison(val):
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    blocks = []
    n = int(min(free * 0.5, 60e9)) // 4
    for sz in (n // 2, n // 4, n // 8, n // 16):
        blocks.append(torch.full((sz,), val, device="cuda", dtype=torch.float32))
    torch.cuda.synchronize()
    del blocks                                   # back to the caching allocator, contents intact


a0, t0 = fwd()
a1, t1 = fwd()
print("repeat, no poison:      logits equal", bool(torch.equal(a0, a1)), "trans equal", bool(torch.equal(t0, t1)))
for val in (float("nan"), 3.0e4, -1.0e30):
    poison(val)
    a2, t2 = fwd()
    print(f"after poison {val!r:>8}: logits equal {bool(torch.equal(a0, a2))} max|d| {float((a0 - a2).abs().max()):.3e} finite {bool(torch.isfinite(a2).all())}; "
          f"trans equal {bool(torch.equal(t0, t2))}")
for name, kw in (("plain bf16 text stream", dict(plain=True)), ("last-layer row subset", dict(subset=True))):
    s0, _ = fwd(**kw)
    a3, t3 = fwd()
    print(f"after a pass with {name}: default logits equal", bool(torch.equal(a0, a3)), f"max|d| {float((a0 - a3).abs().max()):.3e}")
    poison(float("nan"))
    s1, _ = fwd(**kw)
    print(f"{name} after poison: equal", bool(torch.equal(s0, s1)), f"max|d| {float((s0 - s1).abs().max()):.3e}", "finite", bool(torch.isfinite(s1).all()),
          f"max|d| vs default {float((s1 - a0).abs().max()):.3e}")
