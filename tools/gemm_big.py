"""Dense bf16 GEMM throughput of the C-ABI linear operator on large shapes (run on the GPU box): where the tile kernel stands
against the guide's ladder (128^2 tile, two barriers per k-step: 912 TFLOP/s at 4096^3)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import satrn_amd
lib = satrn_amd._lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def bench(fn, iters=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


SHAPES = [(4096, 4096, 4096), (9216, 2048, 512), (9216, 512, 2048), (9216, 1536, 512), (36864, 1024, 256), (147456, 512, 128), (147456, 128, 512)]
if os.environ.get('SATRN_SHAPES'):   # the large dense products of the EfficientSATRN step (1x1 convolutions of the fused-MBConv stages, decoder)
    SHAPES = [(98304, 192, 48), (98304, 48, 192), (98304, 96, 48), (24576, 256, 64), (24576, 64, 256), (6144, 960, 160), (6144, 512, 128), (4096, 1024, 256), (4096, 256, 1024), (4096, 768, 256), (2304, 4096, 1024), (2304, 1024, 4096), (36864, 256, 1024), (147456, 384, 128)]
if os.environ.get('LATE_SHAPES'):    # small-grid products of the late backbone stages, the encoder and the decoder
    SHAPES = [(1536, 1536, 256), (1536, 256, 1536), (6144, 960, 160), (6144, 160, 960), (6144, 512, 128), (6144, 128, 512), (1536, 1536, 512), (1536, 512, 512), (1536, 512, 1536), (4096, 768, 256), (4096, 256, 256), (4096, 1024, 256), (4096, 256, 1024), (4096, 245, 256)]
if os.environ.get('SHAPES'):         # SHAPES="9216,1536,384;9216,384,1536"
    SHAPES = [tuple(int(v) for v in t.split(',')) for t in os.environ['SHAPES'].split(';')]
for M, N, K in SHAPES:
    x = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16(); w = (torch.rand(N, K, device="cuda") * 2 - 1).bfloat16()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    if os.environ.get('ACT_FWD'):     # the training-forward form: bias + GELU + act'(u) stored beside it (epilogue kind 2)
        b = torch.rand(N, device="cuda"); d = torch.empty_like(y)
        us = bench(lambda: lib.satrn_linear_act_fwd(1, P(x), P(w), P(b), P(y), P(d), M, N, K, 4, st()))
    elif os.environ.get('BIAS'):
        b = torch.rand(N, device="cuda")
        us = bench(lambda: lib.satrn_linear_fwd(1, P(x), P(w), P(b), P(y), M, N, K, 0, 0, 0.0, None, 0, st()))
    else:
        us = bench(lambda: lib.satrn_linear_fwd(1, P(x), P(w), None, P(y), M, N, K, 0, 0, 0.0, None, 0, st()))
    print(f"M={M:6d} N={N:5d} K={K:5d}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s  knobs={os.environ.get('SATRN_KNOBS', '-')}")
