"""Dense bf16 weight-gradient throughput of the C-ABI operator (run on the GPU box): dW[N][K] += dY^T X over M rows, on the shapes
of the SwinTRN and EfficientSATRN steps.  SATRN_KNOBS=wgrad_blocks=N sets the grid target (default 96: the partial grid of the side stream)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import satrn_amd
lib = satrn_amd._lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def bench(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


SHAPES = [(9216, 2048, 512), (9216, 512, 2048), (9216, 1536, 512), (9216, 512, 512), (36864, 1024, 256), (36864, 256, 1024), (36864, 768, 256),
          (147456, 512, 128), (147456, 128, 512), (147456, 384, 128), (2304, 4096, 1024), (2304, 1024, 4096), (4096, 1024, 256), (6144, 960, 160), (1536, 1536, 256)]
if os.environ.get('SHAPES'):         # SHAPES="9216,1536,384;9216,384,1536"
    SHAPES = [tuple(int(v) for v in t.split(',')) for t in os.environ['SHAPES'].split(';')]
WITH_BIAS = bool(os.environ.get('WITH_BIAS'))   # also the bias gradient (the form the training step's linears take)
for M, N, K in SHAPES:
    x = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16(); dy = (torch.rand(M, N, device="cuda") * 2 - 1).bfloat16()
    dw = torch.zeros(N, K, device="cuda")
    db = torch.zeros(N, device="cuda") if WITH_BIAS else None
    if os.environ.get('COLD'):   # rotate through operand sets larger than the 256 MB memory-side cache: operands come from HBM, as in a training step
        nset = max(2, int(600e6 / ((M * K + M * N) * 2)) + 1)
        xs = [x] + [x.clone() for _ in range(nset - 1)]; dys = [dy] + [dy.clone() for _ in range(nset - 1)]
        it = [0]
        def fn():
            i = it[0] % nset; it[0] += 1
            lib.satrn_linear_bwd_weight(1, P(dys[i]), N, P(xs[i]), P(dw), P(db) if WITH_BIAS else None, M, N, K, st())
        us = bench(fn)
        print(f"M={M:6d} N={N:5d} K={K:5d}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s  cold ({nset} operand sets)")
        continue
    us = bench(lambda: lib.satrn_linear_bwd_weight(1, P(dy), N, P(x), P(dw), P(db) if WITH_BIAS else None, M, N, K, st()))
    print(f"M={M:6d} N={N:5d} K={K:5d}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s  knobs={os.environ.get('SATRN_KNOBS', '-')} ms128={'off' if os.environ.get('SATRN_WGRAD_NO_MS128') else 'on'}")
