"""How fast does this GPU WRITE?  torch fill_ / copy_ over a rotation of buffers (cold) and over one buffer (cache-resident), 37.7 MB each
(the data-gradient output of a stage-1 projection)."""
import torch
def bench(fn, iters=60):
    for i in range(6): fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
n = 98304 * 192
for nset in (1, 4, 16):
    bufs = [torch.empty(n, device="cuda", dtype=torch.bfloat16) for _ in range(nset)]
    src = [torch.randn(n, device="cuda").bfloat16() for _ in range(nset)]
    us = bench(lambda i: bufs[i % nset].fill_(1.0))
    print(f"fill  {nset:2d} buffers of {n * 2 / 1e6:.1f} MB: {us:6.1f} us  {n * 2 / us / 1e3:7.1f} GB/s written")
    us = bench(lambda i: bufs[i % nset].copy_(src[(i + 1) % nset]))
    print(f"copy  {nset:2d} buffers: {us:6.1f} us  {n * 2 / us / 1e3:7.1f} GB/s written + as much read")
