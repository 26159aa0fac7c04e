"""DDP-vs-single-process gradient equality (semantics of lit_models/base.py:86-91 under PL-DDP): two ranks
launched by torch.distributed.run as the driver does, both on device 0 with gloo moving the CUDA tensors (RCCL refuses two ranks
on one device); the checks themselves live in tests/_ddp_equiv_worker.py."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_allreduced_gradient_equals_single_process_gradient():
    env = dict(os.environ, MART_DIST_BACKEND="gloo", MART_DEVICE_INDEX="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "tests", "_ddp_equiv_worker.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    print("\n", d)
    assert d["master_spread"] == 0.0 and d["shadow_spread"] == 0.0, "rank 0's weights were not broadcast"
    assert d["seed_rank_mixed"] and d["train_mode_outputs_differ_across_ranks"], "ranks draw identical dropout masks"
    assert d["finetune_rel"] < 1e-3, "all-reduced mean gradient != gradient of the concatenated batch"
    assert abs(d["finetune_loss_mean_of_ranks"] - d["finetune_loss_global"]) < 1e-4
    assert d["pretrain_rel"] < 1e-3, "pre-train: mean over ranks of per-rank sub-means not reproduced"
    assert d["pretrain_vs_global_rel"] > 1e-3          # and it is NOT the global mean when the sub-batches are unequal (what PL-DDP does too)
