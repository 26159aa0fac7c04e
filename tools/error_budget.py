"""Per-component bf16 error budget of the mask-row logits (docs/LAB_r01-r05.md section 5).

Runs the fp32-accurate forward (engine_precise) on the golden batch G7 with ONE component class at a time degraded to bf16 operands
(``PreciseUnimoForward.degrade``) and prints how far the mask logits move from the reference's (tests/golden/g7_bench_*.npz), next to
the reference's own one-rounding control (``ctl::``: fp32 math, weight matrices rounded to bf16) and the real bf16 training path.

    python tools/error_budget.py [g7_bench_cond|g7_bench_plain]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_parity_full_gpu as T  # noqa: E402  (golden loader + product builder of the parity tests)

TAGS = ["vis_lin", "vis_attn", "txt_lin_lo", "txt_lin_hi", "txt_attn", "fusion", "head_t", "head_s"]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "g7_bench_cond"
    keys = ("input_ids", "attention_mask", "token_type_ids", "pixel_values", "sep_idx")
    if tag == "bench":
        # the network bench.py times (torch-initialised N(0,0.02) weights, 11292-entity head), first 64 examples of its batch; no reference
        # run exists for it, so the fp32-accurate path (held to the reference at 1e-3 elsewhere) is the yardstick
        import bench
        from mkg_analogy_amd import data_synth as D
        model, lit, cfg = bench.build(16, seed=0, device=torch.device("cuda", 0), entity_head=D.N_ENT)
        model.finalize()
        full = D.make_batch(256, 64, seed=1234, n_labels=D.N_ENT)
        B = 64
        gb = {k: v[:B].cuda() for k, v in full.items()}
        g = {}
        ids = torch.tensor(cfg["analogy_entity_ids"], device="cuda")
        ar = torch.arange(B, device="cuda")
        _, mi = (gb["input_ids"] == 103).nonzero(as_tuple=True)
        rows = mi[:, None]
        model.eval()
        model.set_precision("fp32")
        with torch.no_grad():
            out, _ = model(**{k: gb[k] for k in keys}, return_dict=True)
            ref = out.logits[ar, rows[:, 0]][:, ids].float().cpu()
        model.set_precision("bf16")
    else:
        g = T._load(tag)
        model, lit, cfg = T._product(g)
        batch = T._batch(g)
        B = int(g["B"])
        gb = {k: v.cuda() for k, v in batch.items()}
        ids = torch.tensor(cfg["analogy_entity_ids"], device="cuda")
        ar = torch.arange(B, device="cuda")
        rows = torch.from_numpy(g["trans_row_index"]).cuda()
        ref = torch.from_numpy(g["mask_logits"])
    model.eval()

    def forward():
        with torch.no_grad():
            out, trans = model(**{k: gb[k] for k in keys}, return_dict=True)
            return out.logits[ar, rows[:, 0]][:, ids].float().cpu()

    def report(name, ml):
        d = ml - ref
        print(f"{name:34s} max|dlogit| {float(d.abs().max()):.3e}   rms {float(d.pow(2).mean().sqrt()):.3e}", flush=True)
        return float(d.pow(2).mean().sqrt())

    print(f"{tag}: logit scale {float(ref.abs().max()):.2f}, {ref.numel()} logits")
    if "ctl::mask_logits" in g:
        report("reference control (bf16 weights)", torch.from_numpy(g["ctl::mask_logits"]))
    report("bf16 path (text stream fp16)", forward())
    model.engine.text_f16 = False
    report("bf16 path, plain bf16 text", forward())
    model.engine.text_f16 = True
    model.set_precision("fp32")
    forward()
    pr = model._precise
    base = report("fp32-accurate path", forward())
    var = {}
    for t in TAGS:
        pr.degrade = {t}
        var[t] = report("  only " + t + " in bf16", forward()) ** 2 - base ** 2
    pr.degrade = set(TAGS)
    tot = report("  all components in bf16", forward())
    pr.degrade = set(TAGS) - {"head_t", "head_s"}
    report("  all but the head", forward())
    pr.degrade = set(TAGS) - {"head_t", "head_s", "fusion"}
    report("  all but head + fusion", forward())
    pr.degrade = set(TAGS) - {"head_t", "head_s", "fusion", "txt_lin_hi", "txt_attn"}
    report("  all but head/fusion/txt_hi/attn", forward())
    pr.degrade = {"vis_lin", "vis_attn"}
    report("  vision side only in bf16", forward())
    pr.degrade = set()
    s = sum(max(v, 0.0) for v in var.values())
    print("variance shares (independent-error model): " + ", ".join(f"{t} {100 * max(v, 0) / s:.0f}%" for t, v in var.items()))
    print(f"sqrt(sum of the parts) {s ** 0.5:.3e} vs all-in-bf16 {tot:.3e}")
    model.set_precision("bf16")


if __name__ == "__main__":
    main()
