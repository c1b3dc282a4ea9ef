"""Time the dense GEMM / wgrad entry points of the C-ABI on the shapes the EfficientSATRN step launches most."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import satrn_amd
lib = satrn_amd._lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

def bench(fn, iters=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3  # us

shapes = [(1536, 256, 1536), (1536, 512, 512), (6144, 160, 960), (6144, 128, 512), (1536, 256, 512), (1536, 512, 256)] if os.environ.get('SMALLN') == '1' else [(1536, 1536, 256), (6144, 960, 160), (6144, 512, 128), (1536, 1536, 512), (4096, 1024, 256), (6144, 768, 192)] if os.environ.get('SMALLN') == '2' else [(1536, 1536, 256), (1536, 256, 1536), (6144, 960, 160), (6144, 160, 960), (6144, 512, 128), (6144, 128, 512),
          (4096, 1024, 256), (4096, 256, 1024), (4096, 768, 256), (1536, 512, 512), (98304, 48, 192)]
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
for M, N, K in shapes:
    x = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    dy = torch.randn(M, N, device="cuda").bfloat16(); dw = torch.zeros(N, K, device="cuda")
    stats = torch.zeros(2 * N, device="cuda")
    if which == "fwd":
        us = bench(lambda: lib.satrn_linear_fwd(1, P(x), P(w), None, P(y), M, N, K, 0, 0, 0.0, None, 0, st()))
    else:
        us = bench(lambda: lib.satrn_linear_bwd_weight(1, P(dy), N, P(x), P(dw), None, M, N, K, st()))
    print(f"{which} M={M:6d} N={N:5d} K={K:5d}: {us:8.1f} us  {2.0*M*N*K/us/1e6:8.1f} TFLOP/s  knobs={os.environ.get('SATRN_KNOBS', '-')}")
