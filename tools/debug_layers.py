"""Per-layer drift of the HIP path (bf16) vs the CPU oracle (fp32).  Debug aid, GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import mkgformer_oracle as O
from tests.test_model_gpu import _product, _oracle_sd, BASE, NE, NR
from mkg_analogy_amd import data_synth as D

patch = int(os.environ.get("PATCH", 32)); B = int(os.environ.get("B", 2))
model, lit, cfg, vc = _product(patch, seed=3)
sd = _oracle_sd(vc, 3, cfg["analogy_relation_ids"])
tc = O.TextCfg(vocab_size=BASE + NE + NR + 1)
batch = D.make_batch(B, 64, seed=11)
taps = {}
with torch.no_grad():
    vis = O.vision_embed(sd, vc, batch["pixel_values"]); txt = O.text_embed(sd, tc, batch["input_ids"], batch["token_type_ids"], False)
    taps["vis_emb"], taps["txt_emb"] = vis, txt
    _, trans = O.forward(sd, vc, tc, batch["input_ids"], batch["attention_mask"], batch["token_type_ids"], batch["pixel_values"], batch["sep_idx"], taps=taps)
model.eval()
gb = {k: v.cuda() for k, v in batch.items()}
model.finalize(); model.engine.taps = {}
with torch.no_grad():
    out, tr = model(input_ids=gb["input_ids"], attention_mask=gb["attention_mask"], token_type_ids=gb["token_type_ids"], pixel_values=gb["pixel_values"], sep_idx=gb["sep_idx"], return_dict=True)
ht = model.engine.taps
for k in ["vis_emb", "txt_emb"] + [f"{s}{l}" for l in range(12) for s in ("vis", "txt")]:
    g, r = ht[k].float().cpu(), taps[k]
    print(f"{k:8s} rel-L2 {((g-r).norm()/r.norm()).item():.3e}  max|err| {(g-r).abs().max().item():.3e}  ref rms {r.pow(2).mean().sqrt().item():.3e}")
print("trans rel", ((tr.float().cpu()-trans).norm()/trans.norm()).item())
