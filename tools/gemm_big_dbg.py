"""Timing ablations of the persistent GEMM (wrong results by construction): where a k-step's time goes.
ZERO=1: zero-filled operands (the chip holds a higher clock on zeros: separates DVFS from structure)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import satrn_amd
from satrn_amd import switches as sw
from tools.gemm_big_check import bench, lib, P, st
sw.knob("gemm_big", "2")
for M, N, K in [(9216, 2048, 512), (4096, 4096, 4096), (147456, 128, 128)]:
    for zero in (0, 1):
        x = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16(); w = (torch.rand(N, K, device="cuda") * 2 - 1).bfloat16()
        if zero:
            x.zero_(); w.zero_()
        y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for mt in ("3", "4"):
            sw.knob("gemm_big_mt", mt)
            line = f"M={M} N={N} K={K} MT={mt} zero={zero}:"
            for dbg, name in ((0, "full"), (1, "no-epi"), (4, "no-mfma"), (12, "no-mfma,no-frag-reads"), (13, "dma+barriers only")):
                os.environ["SATRN_TIMING"] = "big_dbg=" + str(str(dbg))
                us = bench(lambda: lib.satrn_linear_fwd(1, P(x), P(w), None, P(y), M, N, K, 0, 0, 0.0, None, 0, st()), iters=30)
                line += f"  {name} {us:.1f}"
            print(line, flush=True)
