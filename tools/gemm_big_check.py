"""Correctness + throughput of the persistent direct-to-LDS GEMM (kernels_gemm_big.hip) through the C-ABI linear operator
(run on the GPU box).  SATRN_KNOBS=gemm_big=2 makes every shape that fits take the kernel; gemm_big=0 is the 4-wave tile kernel."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import satrn_amd
from satrn_amd import switches as sw
lib = satrn_amd._lib.load()
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def bench(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def run(M, N, K, bias, act, mode):
    sw.knob("gemm_big", str(mode))
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = (torch.rand(M, K, device="cuda", generator=g) * 2 - 1).bfloat16(); w = (torch.rand(N, K, device="cuda", generator=g) * 2 - 1).bfloat16()
    b = (torch.rand(N, device="cuda", generator=g) * 2 - 1) if bias else None
    y = torch.full((M, N), 7.0, device="cuda", dtype=torch.bfloat16)
    rc = lib.satrn_linear_fwd(1, P(x), P(w), P(b), P(y), M, N, K, act, 0, 0.0, None, 0, st())
    assert rc == 0
    torch.cuda.synchronize()
    return x, w, b, y


if __name__ == "__main__":
    ok = True
    if "--check" in sys.argv or len(sys.argv) == 1:
        for (M, N, K, bias, act) in [(256, 128, 64, False, 0), (256, 128, 128, True, 0), (300, 136, 192, True, 1), (1000, 384, 512, True, 4), (9216, 2048, 512, True, 4),
                                     (777, 128, 64, False, 0), (6144, 960, 160, True, 0), (1000, 48, 96, False, 0), (3000, 192, 48, True, 1), (555, 256, 24, False, 0), (2304, 1024, 4096, True, 0), (4099, 520, 256, True, 2), (147456, 128, 128, True, 0), (64, 128, 64, False, 0)]:
            x, w, b, y = run(M, N, K, bias, act, 2)
            ref = x.float() @ w.float().t()
            if b is not None: ref = ref + b
            if act == 1: ref = torch.relu(ref)
            if act == 2: ref = ref * torch.sigmoid(ref)
            if act == 4: ref = torch.nn.functional.gelu(ref)
            err = (y.float() - ref).abs().max().item()
            scale = ref.abs().max().item()
            bad = (y.float() - ref).abs() > 2e-2 * scale
            good = err < 1.2e-2 * scale
            ok = ok and good
            print(f"M={M:6d} N={N:5d} K={K:5d} bias={int(bias)} act={act}: max err {err:.3e} (scale {scale:.2f}) {'ok' if good else 'FAIL'} bad={int(bad.sum())}", flush=True)
            if not good:
                idx = bad.nonzero()[:8]
                print("   first bad:", idx.tolist())
    if "--perf" in sys.argv or len(sys.argv) == 1:
        SHAPES = [(9216, 2048, 512), (9216, 512, 2048), (9216, 1536, 512), (9216, 512, 512), (36864, 1024, 256), (36864, 256, 1024), (36864, 768, 256), (147456, 512, 128), (147456, 128, 512),
                  (147456, 384, 128), (147456, 128, 128), (2304, 4096, 1024), (2304, 1024, 4096), (2304, 3072, 1024), (4096, 4096, 4096), (8192, 8192, 8192),
                  (24576, 256, 64), (24576, 64, 256), (6144, 960, 160), (4096, 1024, 256)]
        if os.environ.get("EFF_SHAPES"):
            SHAPES = [(98304, 192, 48), (98304, 48, 192), (98304, 96, 48), (98304, 48, 96), (24576, 256, 64), (24576, 64, 256), (6144, 512, 128), (6144, 128, 512), (6144, 960, 160), (6144, 160, 960),
                      (1536, 1536, 256), (1536, 256, 1536), (1536, 512, 256), (1536, 1536, 512), (1536, 512, 512), (1536, 512, 1536), (4096, 768, 256), (4096, 256, 256), (4096, 1024, 256), (4096, 256, 1024)]
        for M, N, K in SHAPES:
            x = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16(); w = (torch.rand(N, K, device="cuda") * 2 - 1).bfloat16()
            y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            line = f"M={M:6d} N={N:5d} K={K:5d}:"
            for mode in ("0", "2"):
                sw.knob("gemm_big", mode)
                us = bench(lambda: lib.satrn_linear_fwd(1, P(x), P(w), None, P(y), M, N, K, 0, 0, 0.0, None, 0, st()))
                gb = (M * K + N * K + M * N) * 2 / us / 1e3
                line += f"  [{'old' if mode == '0' else 'big'}] {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TF {gb:6.0f} GB/s"
            if (M, N, K) in ((9216, 2048, 512), (147456, 512, 128)):
                b = torch.rand(N, device="cuda")
                for mode in ("0", "2"):
                    sw.knob("gemm_big", mode)
                    us = bench(lambda: lib.satrn_linear_fwd(1, P(x), P(w), P(b), P(y), M, N, K, 4, 0, 0.0, None, 0, st()))
                    line += f"  [{'old' if mode == '0' else 'big'} +bias+GELU] {us:8.1f} us"
            print(line, flush=True)
    sys.exit(0 if ok else 1)
