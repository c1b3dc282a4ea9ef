"""The tall, thin dense products of the fused-MBConv stages through the C-ABI (satrn_linear_fwd_stats): row-streaming kernel
(SATRN_KNOBS=gemm_tall=2) against the tile kernel (gemm_tall=0), cold operands (a rotation of buffer sets larger than the Infinity Cache).
GPU box:  python tools/gemm_tall_bench.py  -> one line per shape and kernel: us, algorithmic GB/s"""
import ctypes, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import satrn_amd
from satrn_amd import switches as sw
lib = satrn_amd._lib.load()
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
# (M, N, K, bnb): projection forward (statistics epilogue) and projection data gradient (BatchNorm-backward sums epilogue)
SHAPES = [(98304, 48, 192, 0), (98304, 192, 48, 2), (24576, 64, 256, 0), (24576, 256, 64, 2), (98304, 192, 48, 0)]
for M, N, K, bnb in SHAPES:
    nset = max(2, int(600e6 // ((M * K + 2 * M * N) * 2)))   # ~600 MB of operand sets: every call reads cold memory
    xs = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(nset)]
    ys = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(nset)]
    bys = [torch.randn(M, N, device="cuda").bfloat16() for _ in range(nset)] if bnb else [None] * nset
    w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
    ss, mr = torch.rand(2 * N, device="cuda") + 0.5, torch.rand(2 * N, device="cuda") + 0.5
    rep = 16 if (N <= 64 and M >= 65536) else 4
    stats = torch.zeros(rep * 2 * N, device="cuda")
    byt = (M * K + M * N * (2 if bnb else 1) + N * K) * 2
    for mode in ("0", "2"):
        sw.knob("gemm_tall", mode)
        sw.knob("gemm_big", "0")
        def call(i):
            lib.satrn_linear_fwd_stats(1, P(xs[i % nset]), P(w), P(ys[i % nset]), M, N, K, P(stats), rep, P(bys[i % nset]), P(ss) if bnb else None, P(mr) if bnb else None, bnb, 0, st())
        for i in range(6): call(i)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 60
        a.record()
        for i in range(iters): call(i)
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / iters * 1e3
        print(f"M={M:6d} N={N:4d} K={K:4d} {'bnb ' if bnb else 'stat'} {'row-streaming' if mode == '2' else 'tile kernel  '}: {us:7.1f} us  {byt / us / 1e3:7.1f} GB/s algorithmic ({byt / 1e6:.1f} MB)", flush=True)
