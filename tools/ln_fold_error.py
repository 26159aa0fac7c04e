"""N1 "by algebra" priced in error: LN(x) W^T + b  ==  rstd * (x (gamma o W)^T - mean * s) + b',  s[n] = sum_k (gamma o W)[n, k],  b' = W beta + b.

For every vision layer of the golden network (G7, conditioned or plain weights; the f32 residual stream tapped from the HIP engine's own forward pass)
and both pre-LN projections (LayerNorm1 -> QKV, LayerNorm2 -> fc1) this prints the error of
  * the shipped operand pair  bf16(LN(x)) x bf16(W)                      (f32 accumulate), and of
  * the folded pair           bf16(x) x bf16(gamma o W), rstd / mean applied to the f32 result
against the f64 product, relative to the rms of the result; plus what the residual stream looks like (|mean| / std per row, largest |x| / std).

    python tools/ln_fold_error.py [g7_bench_cond|g7_bench_plain]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_parity_full_gpu as T  # noqa: E402


def rb(t):
    return t.to(torch.bfloat16).to(torch.float64)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "g7_bench_cond"
    g = T._load(tag)
    model, lit, cfg = T._product(g)
    gb = {k: v.cuda() for k, v in T._batch(g).items()}
    keys = ("input_ids", "attention_mask", "token_type_ids", "pixel_values", "sep_idx")
    eng = model.engine
    eng.taps = {}
    model.eval()
    with torch.no_grad():
        model(**{k: gb[k] for k in keys}, return_dict=True)
    torch.cuda.synchronize()
    taps = {k: v.double() for k, v in eng.taps.items()}
    eng.taps = None
    sd = {k: v.detach().double().cuda() for k, v in model.state_dict().items()}
    H = 768
    print(f"{tag}: vision tower, {taps['vis_emb'].shape[0]} examples x {taps['vis_emb'].shape[1]} tokens")
    print("layer product | row |mean|/std (median, max) | max |x|/std | shipped: rms err, max err | folded: rms err, max err | folded / shipped (rms)")
    for l in range(12):
        x1 = (taps["vis_emb"] if l == 0 else taps[f"vis{l - 1}"]).reshape(-1, H)
        v = f"unimo.encoder.vision_layers.{l}."
        Wqkv = torch.cat([sd[v + f"self_attn.{n}.weight"] for n in ("q_proj", "k_proj", "v_proj")], 0)
        bqkv = torch.cat([sd[v + f"self_attn.{n}.bias"] for n in ("q_proj", "k_proj", "v_proj")], 0)
        # the input of LayerNorm2 is x1 + attention block output; the engine taps only layer boundaries, so LN2 is priced on the same stream (its
        # statistics differ by one residual branch: the same order of magnitude)
        for name, W, b, ln in (("ln1->qkv", Wqkv, bqkv, "layer_norm1"), ("ln2->fc1", sd[v + "mlp.fc1.weight"], sd[v + "mlp.fc1.bias"], "layer_norm2")):
            gam, bet = sd[v + ln + ".weight"], sd[v + ln + ".bias"]
            mean = x1.mean(-1, keepdim=True)
            var = ((x1 - mean) ** 2).mean(-1, keepdim=True)
            rstd = (var + 1e-5).rsqrt()
            lnx = (x1 - mean) * rstd * gam + bet
            ref = lnx @ W.T + b
            ship = rb(lnx) @ rb(W).T + b
            gW = rb(gam[None, :] * W)
            s = gW.sum(-1)
            fold = rstd * (rb(x1) @ gW.T - mean * s[None, :]) + (W @ bet + b)
            sc = ref.pow(2).mean().sqrt()
            e1, e2 = (ship - ref), (fold - ref)
            r = (mean.abs() * rstd).flatten()
            print(f"{l:2d} {name:9s} | {float(r.median()):.3f} {float(r.max()):.3f} | {float((x1.abs() * rstd).max()):6.1f} | "
                  f"{float(e1.pow(2).mean().sqrt() / sc):.2e} {float(e1.abs().max() / sc):.2e} | {float(e2.pow(2).mean().sqrt() / sc):.2e} {float(e2.abs().max() / sc):.2e} | "
                  f"x{float(e2.pow(2).mean().sqrt() / e1.pow(2).mean().sqrt()):.2f}")


if __name__ == "__main__":
    main()
