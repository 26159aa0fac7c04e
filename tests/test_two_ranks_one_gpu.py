"""The N > 1 control flow of bench.py on a 1-GPU box: two ranks launched by torch.distributed.run exactly as the driver does,
both on device 0, gloo moving the CUDA tensors (RCCL refuses two ranks on one device: 'Duplicate GPU detected').  Covers the
rendezvous, the bucketed all-reduce released by the backward pass, the AdamW ranges chained behind the buckets, the barrier +
max-over-ranks timing, the rank gather of the evaluation pass and the one-JSON-line contract; the RCCL transport itself is
covered by tests/test_ddp_nccl_gpu.py (1-rank group) and by the driver's multi-GPU runs."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_share_one_gpu():
    env = dict(os.environ, MART_DIST_BACKEND="gloo", MART_DEVICE_INDEX="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "32",
           "--patch", "32", "--no-cpu-baseline", "--no-kernel-timing"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 64 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and d["loss"] == d["loss"]
    assert d["replica_param_checksum_spread"] == 0.0, d
    # communication diagnostics of the N > 1 path
    assert d["rccl_world"] == 2 and d["comm"]["buckets"] >= 1 and d["comm"]["bucket_dtype"] == "fp32"
    assert d["comm"]["comm_ms_per_step"] > 0 and 0 <= d["comm_exposed_ms"] <= d["comm"]["comm_ms_per_step"] + 1e-6


@pytest.mark.gpu
def test_bench_launches_its_own_ranks():
    """``python bench.py --gpus 2`` as typed (no WORLD_SIZE / RANK in the environment: the shape of the driver's N = 1 command with a larger N) starts its
    own two ranks under torch.distributed.run and relays rank 0's single JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK")}
    env.update(MART_DIST_BACKEND="gloo", MART_DEVICE_INDEX="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "16", "--patch", "32",
           "--no-cpu-baseline", "--no-kernel-timing", "--train-only"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 32 and d["value"] > 0
    assert d["replica_param_checksum_spread"] == 0.0, d
