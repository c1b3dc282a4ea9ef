"""Data-parallel helpers (new: the reference has no multi-GPU path).  One process per GPU; the model is a full
replica; gradients live in ONE flat fp32 buffer, so the exchange step is a single (optionally bucketed) all-reduce.
Backend "nccl" is RCCL on ROCm; the same code runs over "gloo" on CPU tensors (used by the CPU tests)."""
import random

import torch
import torch.distributed as dist


def bucket_bounds(numel, bucket_elems):
    """[(start, end)) slices of a flat buffer, every bucket a multiple of 4 elements except the last."""
    bucket_elems = max(4, (int(bucket_elems) // 4) * 4)
    out, s = [], 0
    while s < numel:
        e = min(numel, s + bucket_elems)
        out.append((s, e))
        s = e
    return out


def allreduce_flat(flat, bucket_elems=None, average=False, group=None):
    """Sum (or mean) `flat` over the ranks in place.  bucket_elems=None -> one collective (xGMI all-reduce of the whole
    108.9 MB EfficientSATRN gradient is ~1 ms); otherwise one async collective per bucket, waited at the end."""
    world = dist.get_world_size(group)
    if world == 1:
        return flat
    if bucket_elems is None:
        dist.all_reduce(flat, group=group)
    else:
        works = [dist.all_reduce(flat[s:e], group=group, async_op=True) for s, e in bucket_bounds(flat.numel(), bucket_elems)]
        for w in works:
            w.wait()
    if average:
        flat.mul_(1.0 / world)
    return flat


def broadcast_flat(flat, src=0, group=None):
    if dist.get_world_size(group) > 1:
        dist.broadcast(flat, src, group=group)
    return flat


def shard_batch(n_items, rank, world):
    """contiguous shard [start, end) of a global batch for this rank (independent samples: no data-path collective)"""
    per = (n_items + world - 1) // world
    s = min(n_items, rank * per)
    return s, min(n_items, s + per)


def sync_bn_buffers(model, mode="broadcast", src=0, group=None):
    """BatchNorm running statistics at checkpoint time.  Every rank normalises with its OWN batch statistics (what
    N independent reference processes would compute: no SyncBN), so the running_mean / running_var buffers drift apart between
    ranks while the parameters stay identical.  Before a state_dict is written they are made equal:
      mode "broadcast": rank `src`'s buffers everywhere (a checkpoint equals what a 1-GPU run on rank src's shard would store);
      mode "average":   running_mean / running_var averaged over the ranks (an unbiased estimate over the GLOBAL batch stream);
                        every other buffer (num_batches_tracked, SwinTRN's index / mask tables) is broadcast.
    Works on the model's flat buffers (one or two collectives, not one per BatchNorm)."""
    if mode not in ("broadcast", "average"):
        raise ValueError("mode must be 'broadcast' or 'average'")
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    f32, i64 = model._flat[1], model._flat[2]
    if world == 1:
        return model
    if mode == "average":
        mean = f32.clone()
        dist.all_reduce(mean, group=group)
        mean.mul_(1.0 / world)
    dist.broadcast(f32, src, group=group)
    dist.broadcast(i64, src, group=group)
    if mode == "average":
        for name, kind, shp, off, numel, node, leaf in model._entries:
            if kind == 1 and leaf in ("running_mean", "running_var"):
                f32[off: off + numel].copy_(mean[off: off + numel])
    return model


def state_dict(model, bn_mode="broadcast", src=0, group=None):
    """model.state_dict() of a data-parallel replica: BatchNorm buffers synchronised first (sync_bn_buffers), so that every rank
    would write the same checkpoint (the reference saves `model.state_dict()`, train_modules/train_single_opt.py:497-512).
    The rank-shared coin is not part of it (the reference's state_dict has no such key): checkpoint it beside the model with
    coin.state_dict() / coin.load_state_dict()."""
    sync_bn_buffers(model, bn_mode, src, group)
    return model.state_dict()


class SharedCoin:
    """The teacher-forcing coin of networks/EfficientSATRN.py:489 (`random.random() < teacher_forcing_ratio`, one flip per batch)
    for data-parallel ranks: every rank must take the SAME decoder branch in a step, or their gradient exchanges would pair a
    teacher-forced backward with an autoregressive one.  All ranks build it with the same seed (the reference seeds Python's
    `random` in set_seed, utils/utils.py:167-171) and flip it once per step."""

    def __init__(self, seed=21):
        self._rng = random.Random(int(seed))
        self.flips = 0

    def random(self):
        self.flips += 1
        return self._rng.random()

    def teacher_forced(self, ratio):
        return self.random() < ratio

    def state_dict(self):
        """generator state + flip count: a resumed run continues the SAME branch sequence on every rank"""
        return {"rng": self._rng.getstate(), "flips": self.flips}

    def load_state_dict(self, sd):
        st = sd["rng"]
        self._rng.setstate((st[0], tuple(st[1]), st[2]))
        self.flips = int(sd["flips"])

    def check_in_step(self, teacher_forced=None, group=None):
        """debug aid: raise if the ranks disagree on the flip count (a rank skipped or repeated a step: an uneven last shard, a retry)
        or on the branch just taken.  One 3-element all-reduce (MIN and MAX folded into one by sending x and -x): call it every N steps."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return
        tf = -1.0 if teacher_forced is None else float(bool(teacher_forced))
        v = torch.tensor([float(self.flips), -float(self.flips), tf, -tf], dtype=torch.float64)
        if dist.get_backend(group) == "nccl":
            v = v.cuda()
        dist.all_reduce(v, op=dist.ReduceOp.MAX, group=group)
        v = v.cpu()
        if v[0] != -v[1] or v[2] != -v[3]:
            raise RuntimeError(f"SharedCoin drifted between ranks: flips max {int(v[0])} min {int(-v[1])}, branch max {v[2]} min {-v[3]}")


_exposed = None


def last_exchange_exposed_ms():
    """GPU time between the end of the last backward segment and the start of the optimizer of the most recent dp_train_step on
    this rank = the part of the gradient exchange that was NOT hidden behind the backward (synchronises); None if the last step
    had no exchange."""
    if _exposed is None:
        return None
    a, b = _exposed
    b.synchronize()
    return a.elapsed_time(b)


def dp_train_step(model, images, expected, lr, overlap=True, force_exchange=False, **kw):
    """One data-parallel step.  overlap=True (eager execution): the backward is cut after the last backbone stage; the all-reduce
    of everything finished by then (74 % of the flat gradient) is started asynchronously (RCCL on its own stream) and runs
    beside the backward of the early backbone, whose range is reduced at the end.  Otherwise:
    forward/backward -> one flat all-reduce -> clip + AdamW.  Both end with grad_scale = 1/world inside the optimizer.
    force_exchange: take the split-phase path and issue the collectives even when the group has ONE rank (how the one-GPU
    test box executes the RCCL calls and the stream hand-offs of the real path).
    teacher_forcing_ratio=... (in kw): the branch coin is flipped ONCE, by the call that starts the step; every rank must flip the same
    coin (model.set_coin(dp.SharedCoin(seed)) on all ranks), or their exchanges would pair different decoder branches."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1 and not (force_exchange and dist.is_initialized()):
        model.train_step(images, expected, lr, **kw)
        return
    if overlap and not kw.get("use_graph", False):
        kw2 = {k: v for k, v in kw.items() if k != "use_graph"}
        # two exchange buckets: segments 0-2 (decoder, encoder transformer, last backbone stage: 74 % of the gradient) in
        # one engine call, reduced while segment 3 (the early backbone, a third of the backward time) runs; every extra
        # cut costs a side-stream join (measured: four segments +0.33 ms per step, this single cut +0.05 ms)
        works = []
        flat = model.flat_grad()
        model.train_step(images, expected, lr, phase=16 + 0 + 4 * 2, **kw2)
        cut = model.segment_range(2)[0]
        if cut < flat.numel():
            works.append(dist.all_reduce(flat[cut:], async_op=True))
        model.train_step(images, expected, lr, phase=16 + 3, **kw2)
        if cut > 0:
            works.append(dist.all_reduce(flat[:cut], async_op=True))
        ev = _mark(flat)
        for w in works:
            w.wait()
    else:
        model.train_step(images, expected, lr, phase=1, **kw)
        ev = _mark(model.flat_grad())
        allreduce_flat(model.flat_grad())
    _mark_end(ev, model.flat_grad())
    model.train_step(images, expected, lr, phase=2, grad_scale=1.0 / world, **kw)


def _mark(flat):
    global _exposed
    _exposed = None
    if not flat.is_cuda:
        return None
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()   # the backward's last segment has been issued on the current stream
    return ev


def _mark_end(ev, flat):
    global _exposed
    if ev is None:
        return
    end = torch.cuda.Event(enable_timing=True)
    end.record()  # the current stream has been made to wait for every collective: the optimizer starts here
    _exposed = (ev, end)
