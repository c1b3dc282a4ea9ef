"""CPU: the data-parallel exchange step over gloo with 2 ranks (the GPU path uses the same code over RCCL)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import satrn_amd
    from satrn_amd import dp
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 100003
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    ref = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world)) / world
    a = dp.allreduce_flat(g.clone(), average=True)
    b = dp.allreduce_flat(g.clone(), bucket_elems=4099, average=True)
    p = torch.full((17,), float(rank))
    dp.broadcast_flat(p, 0)
    ok = torch.allclose(a, ref) and torch.allclose(b, ref) and bool((p == 0).all())
    s, e = dp.shard_batch(256, rank, world)
    ok = ok and (e - s) == 128 and s == rank * 128
    q.put((rank, ok))
    dist.destroy_process_group()


def test_allreduce_broadcast_shard_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_bucket_bounds_cover_buffer():
    sys.path.insert(0, ROOT)
    from satrn_amd import dp
    b = dp.bucket_bounds(27221141, 4 << 20)
    assert b[0][0] == 0 and b[-1][1] == 27221141
    assert all(x[1] == y[0] for x, y in zip(b, b[1:]))
    assert all((e - s) % 4 == 0 for s, e in b[:-1])
