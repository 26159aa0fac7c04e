"""Worker of tests/test_ddp_equivalence_gpu.py (launched with torch.distributed.run, 2 ranks, both on device 0 over gloo).

Checks, on real kernels:
  1. GradSync broadcasts rank 0's master weights (rank 1 perturbs its own copy first) and refreshes the bf16 shadows;
  2. fine-tune: the all-reduced mean gradient of two 8-example shards == the single-process gradient of the concatenated
     16-example batch (equal shards: mean over ranks of per-rank means == global mean; lit_models/base.py:86-91 semantics);
  3. pre-train: entity / relation sub-batches differ in size per rank, so DDP yields the mean over ranks of
     (entity sub-mean + relation sub-mean) -- compared with the same quantity accumulated in one process;
  4. ranks draw different dropout masks (base_seed mixed with the rank).
Prints one JSON line on rank 0."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    from mkg_analogy_amd import data_synth as D
    from mkg_analogy_amd.distributed import GradSync, init_from_env
    rank, local, world = init_from_env()
    assert world == 2
    dev = torch.device("cuda", local)
    model, lit, cfg = bench.build(32, seed=0, device=dev, backbone="mkgformer")
    model.finalize()
    st = model.store
    if rank == 1:
        st.master.add_(0.01 * torch.randn_like(st.master))            # a replica that started from different weights
        st.refresh_shadows()
    seed0 = int(model.base_seed)
    sync = GradSync(model)
    cs = st.master.double().sum().reshape(1)
    sh = st.shadow.float().double().sum().reshape(1)
    both = [torch.zeros_like(cs) for _ in range(2)]
    dist.all_gather(both, cs)
    both_sh = [torch.zeros_like(sh) for _ in range(2)]
    dist.all_gather(both_sh, sh)
    out = {"master_spread": float((both[0] - both[1]).abs()), "shadow_spread": float((both_sh[0] - both_sh[1]).abs()),
           "seed_rank_mixed": (int(model.base_seed) != seed0) == (rank != 0)}
    eng = model.engine
    model.eval()                                                       # dropout off: the comparison is deterministic

    def ddp_grad(batch, pretrain):
        lit.args.pretrain = int(pretrain)
        st.zero_grad()
        sync.begin()
        eng.grad_ready_async = sync.reducer.ready
        loss = lit.training_step(dict(batch), 1)
        loss.backward()
        sync.finish()
        torch.cuda.synchronize()
        return st.grad.clone() * sync.grad_scale, float(loss.detach())

    def local_grad(batches, pretrain, scale):
        lit.args.pretrain = int(pretrain)
        eng.grad_ready_async = None
        eng.grad_ready = None
        st.zero_grad()
        tot = 0.0
        for b in batches:
            loss = lit.training_step(dict(b), 1)
            loss.backward()
            tot += float(loss.detach())
        torch.cuda.synchronize()
        return st.grad.clone() * scale, tot * scale

    def shard(b, r):
        return {k: v[8 * r:8 * r + 8] for k, v in b.items()}

    # ---- fine-tune: equal shards
    full = D.make_batch(16, 64, seed=5, device=dev)
    g_ddp, l_mine = ddp_grad(shard(full, rank), False)
    g_ref, l_ref = local_grad([full], False, 1.0)
    out["finetune_rel"] = float((g_ddp - g_ref).norm() / g_ref.norm())
    lm = torch.tensor([l_mine], device=dev, dtype=torch.float64)
    dist.all_reduce(lm)
    out["finetune_loss_mean_of_ranks"], out["finetune_loss_global"] = float(lm) / 2, l_ref
    # ---- pre-train: unequal entity / relation sub-batches per rank
    pre = D.make_batch(16, 96, seed=6, device=dev, pretrain=True)
    pt = pre["pre_type"].clone()
    pt[:8] = torch.tensor([1, 1, 1, 1, 1, 1, 2, 2], device=dev)      # rank 0: 6 entity / 2 relation rows
    pt[8:] = torch.tensor([2, 2, 2, 2, 2, 1, 1, 2], device=dev)      # rank 1: 2 entity / 6 relation rows
    lab = torch.where(pt == 2, pre["label"] % D.N_REL, pre["label"])
    pre["pre_type"], pre["label"] = pt, lab
    g_ddp, _ = ddp_grad(shard(pre, rank), True)
    g_ref, _ = local_grad([shard(pre, 0), shard(pre, 1)], True, 0.5)  # mean over ranks of the per-rank (sub-mean + sub-mean)
    g_glob, _ = local_grad([pre], True, 1.0)                          # what a single process on the 16 rows would compute
    out["pretrain_rel"] = float((g_ddp - g_ref).norm() / g_ref.norm())
    out["pretrain_vs_global_rel"] = float((g_ddp - g_glob).norm() / g_glob.norm())   # differs by construction (unequal sub-batches)
    # ---- dropout masks differ across ranks (train mode, same batch)
    model.train()
    lit.args.pretrain = 0
    eng.grad_ready_async = None
    with torch.no_grad():
        _, tr = model(**{k: full[k][:4] for k in ("input_ids", "attention_mask", "token_type_ids", "pixel_values", "sep_idx")}, return_dict=True)
    tsum = tr.double().sum().reshape(1)
    ts = [torch.zeros_like(tsum) for _ in range(2)]
    dist.all_gather(ts, tsum)
    out["train_mode_outputs_differ_across_ranks"] = bool((ts[0] - ts[1]).abs() > 0)
    oks = [None, None]
    dist.all_gather_object(oks, out["seed_rank_mixed"])
    out["seed_rank_mixed"] = all(oks)
    if rank == 0:
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
