"""Throughput of the persistent weight-gradient kernel against the 4-wave one (run on the GPU box)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.gemm_big_check import bench, lib, P, st
SHAPES = [(9216, 2048, 512), (9216, 512, 2048), (9216, 1536, 512), (9216, 512, 512), (36864, 1024, 256), (36864, 256, 1024), (36864, 768, 256), (36864, 256, 256),
          (147456, 512, 128), (147456, 128, 512), (147456, 384, 128), (147456, 128, 128), (2304, 4096, 1024), (2304, 1024, 4096), (2304, 3072, 1024), (2304, 1024, 1024), (4096, 1024, 256), (2048, 512, 512)]
for M, N, K in SHAPES:
    x = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16(); dy = (torch.rand(M, N, device="cuda") * 2 - 1).bfloat16()
    dw = torch.zeros(N, K, device="cuda"); db = torch.zeros(N, device="cuda")
    line = f"M={M:6d} N={N:5d} K={K:5d}:"
    for mode in ("0", "2"):
        sw.knob("wgrad_big", mode)
        us = bench(lambda: lib.satrn_linear_bwd_weight(1, P(dy), N, P(x), P(dw), P(db), M, N, K, st()))
        us2 = bench(lambda: lib.satrn_linear_bwd_weight(1, P(dy), N, P(x), P(dw), None, M, N, K, st()))
        line += f"  [{'old' if mode == '0' else 'big'}] +db {us:7.1f} us {2.0*M*N*K/us/1e6:6.1f} TF | no db {us2:7.1f} us {2.0*M*N*K/us2/1e6:6.1f} TF"
    print(line, flush=True)
