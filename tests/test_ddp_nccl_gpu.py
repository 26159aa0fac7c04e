"""The RCCL code path on a 1-GPU box: bench.py under torch.distributed.run with a 1-rank ``nccl`` group and the bucketed,
backward-overlapped all-reduce forced on (MART_FORCE_PG=1).  A sum over one rank is the identity, so the step must
reproduce the plain single-process run; what this covers is process-group init, the side-stream / event choreography
and the collectives themselves on real hardware (the world-2 semantics are covered by tests/test_ddp_gloo_cpu.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(cmd, env_extra):
    env = dict(os.environ, **env_extra)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=420)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_bench_under_torchrun_nccl_one_rank():
    common = ["bench.py", "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "16", "--patch", "32", "--no-cpu-baseline",
              "--no-kernel-timing"]
    plain = _bench([sys.executable] + common, {})
    ddp = _bench([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                  "--master-port", "29517"] + common, {"MART_FORCE_PG": "1"})
    print("\nplain", plain["loss"], plain["value"], "| nccl(1 rank, forced buckets)", ddp["loss"], ddp["value"])
    assert ddp["n_gpus"] == 1 and ddp["config"]["parallelism"] == "dp1"
    # same seeds, same kernels, and every reduction on the gradient path is ordered (no float atomics): a sum over one rank is the
    # identity, so four train-mode steps through the bucketed RCCL path reproduce the plain run to the last printed digit
    # (with the round-1 atomics four runs of this command spread over 0.028 and the bound had to be 0.1)
    assert ddp["loss"] == plain["loss"], (ddp["loss"], plain["loss"])
