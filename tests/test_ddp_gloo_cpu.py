"""world_size-2 gloo tests of the N>1 path: bucketed gradient all-reduce in backward-completion order and rank gathering."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from mkg_analogy_amd.distributed import BucketedAllReduce, all_gather_ranks, init_from_env
    r, _, w = init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    n = 10_000
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(n, generator=g)
    mine = flat.clone()
    buckets = [(0, 3000), (3000, 6500), (6500, n)]
    red = BucketedAllReduce(flat, buckets)
    red.begin()
    red.ready(2999)                      # nothing complete yet
    assert red.next == 0
    red.ready(3000)
    assert red.next == 1
    red.ready(7000)                      # second bucket complete, third not
    assert red.next == 2
    red.finish()                         # flushes the tail
    assert red.next == 3
    other = torch.randn(n, generator=torch.Generator().manual_seed(100 + (1 - rank)))
    ok = torch.allclose(flat, mine + other, atol=1e-6)
    # bf16 gradient buckets (MART_GRAD_BUCKET_DTYPE=bf16: half the bytes per link) and the communication statistics bench.py reports
    flat2 = mine.clone()
    red2 = BucketedAllReduce(flat2, buckets, bucket_dtype="bf16", timing=True)
    red2.begin()
    red2.ready(6500)
    red2.finish()
    want = mine.to(torch.bfloat16).float() + other.to(torch.bfloat16).float()
    ok = ok and torch.allclose(flat2, want.to(torch.bfloat16).float(), atol=1e-6)
    st = red2.stats()
    ok = ok and set(st) >= {"steps", "comm_ms", "comm_exposed_ms", "buckets", "bucket_mb", "bucket_dtype"} and st["bucket_dtype"] == "bf16" and st["buckets"] == 3
    ranks = all_gather_ranks(np.array([rank * 10 + 1, rank * 10 + 2]))
    q.put((rank, ok, ranks.tolist()))
    dist.destroy_process_group()


def test_bucketed_allreduce_and_rank_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, ranks in res:
        assert ok, f"rank {rank}: reduced gradients differ from the sum"
        assert ranks == [1, 2, 11, 12]


def test_bucket_partition_covers_flat_buffer():
    class _S:
        total = 1000
    from mkg_analogy_amd.params import FlatStore
    b = FlatStore.buckets(_S(), 300)
    assert b == [(0, 300), (300, 600), (600, 900), (900, 1000)]
