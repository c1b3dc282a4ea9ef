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


class _FakeModel:
    """records the call sequence of dp.dp_train_step and lets 'segments' fill their flat-gradient ranges"""

    def __init__(self, rank):
        self.g = torch.zeros(1000)
        self.rank = rank
        self.calls = []
        self.ranges = [(700, 1000), (400, 700), (100, 400), (0, 100)]

    def flat_grad(self):
        return self.g

    def segment_range(self, seg):
        return self.ranges[seg]

    def train_step(self, images, expected, lr, phase=3, grad_scale=1.0, **kw):
        self.calls.append((phase, grad_scale))
        if phase == 1:
            self.g[:] = float(self.rank + 1)
        elif phase >= 16:
            k, k_to = phase & 3, max((phase >> 2) & 3, phase & 3)
            if k == 0:
                self.g.zero_()
            for seg in range(k, k_to + 1):
                lo, hi = self.ranges[seg]
                self.g[lo:hi] = float(self.rank + 1)


def _overlap_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from satrn_amd import dp
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = _FakeModel(rank), _FakeModel(rank)
    dp.dp_train_step(a, None, None, 1e-3, overlap=True)
    dp.dp_train_step(b, None, None, 1e-3, overlap=False)
    ok = bool((a.g == 3.0).all()) and bool((b.g == 3.0).all())  # 1 + 2 summed over the two ranks, every element once
    ok = ok and [c[0] for c in a.calls] == [24, 19, 2] and [c[0] for c in b.calls] == [1, 2]
    ok = ok and a.calls[-1][1] == 0.5 and b.calls[-1][1] == 0.5
    q.put((rank, ok))
    dist.destroy_process_group()


def test_overlapped_exchange_reduces_every_segment_once_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31000 + os.getpid() % 2000
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


class _FakeBNModel:
    """the two flat buffers + entry table dp.sync_bn_buffers works on (networks._SATRNBase has the same attributes)"""

    def __init__(self, rank):
        f32 = torch.zeros(12)
        f32[0:4] = float(rank + 1)          # running_mean
        f32[4:8] = 10.0 * (rank + 1)        # running_var
        f32[8:12] = 7.0                     # a constant table (identical on all ranks)
        i64 = torch.full((2,), 5 + rank, dtype=torch.int64)
        self._flat = [torch.zeros(1), f32, i64]
        self._entries = [("bn.running_mean", 1, (4,), 0, 4, None, "running_mean"), ("bn.running_var", 1, (4,), 4, 4, None, "running_var"),
                         ("tab", 1, (4,), 8, 4, None, "relative_position_index"), ("bn.num_batches_tracked", 2, (2,), 0, 2, None, "num_batches_tracked")]

    def state_dict(self):
        return {"f32": self._flat[1].clone(), "i64": self._flat[2].clone()}


def _bn_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from satrn_amd import dp
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a = _FakeBNModel(rank)
    dp.sync_bn_buffers(a, "broadcast")
    ok = bool((a._flat[1][0:4] == 1.0).all()) and bool((a._flat[1][4:8] == 10.0).all()) and bool((a._flat[2] == 5).all())
    b = _FakeBNModel(rank)
    sdict = dp.state_dict(b, bn_mode="average")
    ok = ok and bool((sdict["f32"][0:4] == 1.5).all()) and bool((sdict["f32"][4:8] == 15.0).all()) and bool((sdict["f32"][8:12] == 7.0).all())
    ok = ok and bool((sdict["i64"] == 5).all())
    # the shared teacher-forcing coin: the same branch on every rank, step after step, and a fair coin at the ratio
    coin = dp.SharedCoin(seed=1234)
    flips = torch.tensor([1.0 if coin.teacher_forced(0.55) else 0.0 for _ in range(200)])
    both = [torch.zeros_like(flips) for _ in range(world)]
    dist.all_gather(both, flips)
    ok = ok and all(torch.equal(both[0], f) for f in both) and 0.4 < flips.mean().item() < 0.7 and coin.flips == 200
    q.put((rank, ok))
    dist.destroy_process_group()


def test_bn_buffer_sync_and_shared_coin_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33000 + os.getpid() % 2000
    procs = [ctx.Process(target=_bn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]
