"""The LayerNorm fold in the model (engine.ln_fold: no_grad forward passes of the bf16 configuration; modeling_unimo.py:509 -> :223-225, :518 -> :284-286):
against the CPU oracle at north_star's bf16 tolerance, against the unfused pass, and never in a pass that keeps activations for a backward pass."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mkgformer_oracle as O  # noqa: E402
from test_model_gpu import BASE, NE, NR, _oracle_sd, _product  # noqa: E402


@pytest.mark.parametrize("patch,B,L", [(16, 2, 64), (32, 3, 37)])
def test_folded_eval_pass_vs_oracle_and_unfused(patch, B, L):
    from mkg_analogy_amd import data_synth as D
    model, lit, cfg, vc = _product(patch, seed=3, conditioned=True)
    sd = _oracle_sd(vc, 3, cfg["analogy_relation_ids"], True)
    tc = O.TextCfg(vocab_size=BASE + NE + NR + 1)
    batch = D.make_batch(B, L, seed=11)
    ids = torch.tensor(cfg["analogy_entity_ids"])
    with torch.no_grad():
        _, trans_ref = O.forward(sd, vc, tc, batch["input_ids"], batch["attention_mask"], batch["token_type_ids"], batch["pixel_values"], batch["sep_idx"], train=False)
        _, ml_ref = O.finetune_loss(sd, trans_ref, batch["input_ids"], batch["label"], batch["rel_idx"], batch["q_head_idx"], batch["a_head_idx"], ids, alpha=0.43)
    model.eval()
    gb = {k: v.cuda() for k, v in batch.items()}
    kw = dict(input_ids=gb["input_ids"], attention_mask=gb["attention_mask"], token_type_ids=gb["token_type_ids"], pixel_values=gb["pixel_values"],
              sep_idx=gb["sep_idx"], return_dict=True)
    _, mi = (gb["input_ids"] == 103).nonzero(as_tuple=True)
    eng = model.engine
    import os
    assert eng.ln_fold == (os.environ.get("MART_LN_FOLD", "0") == "1"), "opt-in: by default a no_grad pass is bit-identical to the forward pass of a training step"
    eng.ln_fold = True
    calls = []
    from mkg_analogy_amd import ops
    real = ops.ln_stats_finalize
    ops.ln_stats_finalize = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        with torch.no_grad():
            out_f, trans_f = model(**kw)
        n_fold = len(calls)
        out_g, trans_g = model(**kw)                       # gradients enabled: activations are kept for a backward pass -> the LayerNorm pass stays
        assert len(calls) == n_fold, "a pass that keeps activations must not fold"
    finally:
        ops.ln_stats_finalize = real
    assert n_fold == 2 * eng.n_layers - 1                  # every vision LayerNorm but layer 0's layer_norm1 (its input comes from the embedding kernels)
    eng.ln_fold = False
    with torch.no_grad():
        out_u, trans_u = model(**kw)
    eng.ln_fold = True
    assert torch.equal(trans_u, trans_g.detach()), "no_grad without the fold == the forward pass of a training step"
    rows = torch.arange(B, device="cuda")
    ml_f, ml_u = out_f.logits[rows, mi][:, ids.cuda()].float().cpu(), out_u.logits[rows, mi][:, ids.cuda()].float().cpu()
    scale = max(1.0, float(ml_ref.abs().max()))
    e_f, e_u = float((ml_f - ml_ref).abs().max()), float((ml_u - ml_ref).abs().max())
    r_f, r_u = float((ml_f - ml_ref).pow(2).mean().sqrt()), float((ml_u - ml_ref).pow(2).mean().sqrt())
    print(f"\npatch {patch} B {B} L {L}: mask logits vs oracle: folded max {e_f:.3e} rms {r_f:.3e}; unfused max {e_u:.3e} rms {r_u:.3e}; folded vs unfused max {float((ml_f - ml_u).abs().max()):.3e}")
    assert e_f < 1e-2 * scale and r_f < 5e-3, "folded pass outside north_star's bf16 tolerance"
    assert r_f < 1.5 * r_u + 5e-4, "the fold must not cost accuracy beyond rounding"
    ev = lit._eval(dict(gb), 0)                             # the bf16 validation step runs under no_grad: folded
    ranks_ref = O.ranks_double_sort(ml_ref, batch["label"])
    amb = ((ml_ref - ml_ref[torch.arange(B), batch["label"]][:, None]).abs() < 2 * e_f).sum(1).numpy() - 1
    assert np.all(np.abs(ev["entity_ranks"] - ranks_ref) <= amb)
    eng.ln_fold = os.environ.get("MART_LN_FOLD", "0") == "1"
