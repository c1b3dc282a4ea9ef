"""The encoder self-attention region of BASELINE configs[1] (networks/EfficientSATRN.py:198-228,265-268: LayerNorm -> fused
q/k/v projection -> scaled-dot-product attention -> out projection, and their backward), alone, through the operator-level
C-ABI, at the benchmark's dims (B=32, 4x12=48 tokens, d=512, 8 heads).  Every dispatch of this process belongs to the region,
so a rocprofv3 --pmc pass over it attributes SQ_VALU_MFMA_BUSY_CYCLES per kernel without guessing which of the training
step's 222 GEMM launches is the QKV projection.

    python tools/enc_attn_region.py [--iters N] [--dtype bf16|f32]         -> one JSON line (HIP-event timing per op)
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d DIR -- python3 tools/enc_attn_region.py --iters 20
    python tools/pmc_mfma.py DIR profiles/r02_mfma_busy.json
"""
import argparse
import ctypes
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    import satrn_amd
    lib = satrn_amd._lib.load()
    dt = 1 if a.dtype == "bf16" else 0
    tdt = torch.bfloat16 if dt else torch.float32
    B, L, D, H = a.batch, 48, 512, 8
    M, hd = B * L, D // H
    dev = "cuda"
    P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(M, D, generator=g)).to(dev).to(tdt)
    lnw, lnb = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    wqkv = (torch.randn(3 * D, D, generator=g) * (2.0 / (2 * D)) ** 0.5).to(dev)
    wo = (torch.randn(D, D, generator=g) * (2.0 / (2 * D)) ** 0.5).to(dev)
    bqkv, bo = torch.zeros(3 * D, device=dev), torch.zeros(D, device=dev)
    pk = lambda w: (torch.empty(w.shape, dtype=tdt, device=dev), torch.empty(w.shape[1], w.shape[0], dtype=tdt, device=dev))
    wqkv_f, wqkv_b = pk(wqkv)
    wo_f, wo_b = pk(wo)
    ok = lambda rc: (_ for _ in ()).throw(RuntimeError(lib.satrn_last_error().decode())) if rc else None
    ok(lib.satrn_pack_dense(dt, P(wqkv), P(wqkv_f), P(wqkv_b), 3 * D, D, 3 * D, st()))
    ok(lib.satrn_pack_dense(dt, P(wo), P(wo_f), P(wo_b), D, D, D, st()))
    y = torch.empty(M, D, dtype=tdt, device=dev)
    mr = torch.empty(2 * M, device=dev)
    qkv = torch.empty(M, 3 * D, dtype=tdt, device=dev)
    att = torch.empty(M, D, dtype=tdt, device=dev)
    lse = torch.empty(B * H * L, device=dev)
    out = torch.empty(M, D, dtype=tdt, device=dev)
    dout = torch.randn(M, D, generator=g).to(dev).to(tdt)
    datt = torch.empty(M, D, dtype=tdt, device=dev)
    dqkv = torch.zeros(M, 3 * D, dtype=tdt, device=dev)
    dy = torch.empty(M, D, dtype=tdt, device=dev)
    dwo, dbo = torch.zeros(D, D, device=dev), torch.zeros(D, device=dev)
    dwqkv, dbqkv = torch.zeros(3 * D, D, device=dev), torch.zeros(3 * D, device=dev)
    ws = torch.zeros(2 * B * H * L * 64, dtype=tdt, device=dev)
    es = qkv.element_size()
    q_, k_, v_ = (ctypes.c_void_p(qkv.data_ptr() + i * D * es) for i in range(3))
    dq_, dk_, dv_ = (ctypes.c_void_p(dqkv.data_ptr() + i * D * es) for i in range(3))
    temp = math.sqrt(D)
    ops = [
        ("layernorm_fwd", 0.0, lambda: lib.satrn_layernorm_fwd(dt, P(x), None, P(lnw), P(lnb), P(y), P(mr), M, D, 1e-5, st())),
        ("qkv_projection_fwd", 2.0 * M * 3 * D * D, lambda: lib.satrn_linear_fwd(dt, P(y), P(wqkv_f), P(bqkv), P(qkv), M, 3 * D, D, 0, 0, 0.0, None, 0, st())),
        ("attention_fwd", 4.0 * B * H * L * L * hd, lambda: lib.satrn_attention_fwd(dt, q_, k_, v_, P(att), P(lse), B, H, L, L, hd, 3 * D, 3 * D, 3 * D, D, 0, None, 0, 2, temp, 0.0, None, 0, st())),
        ("out_projection_fwd", 2.0 * M * D * D, lambda: lib.satrn_linear_fwd(dt, P(att), P(wo_f), P(bo), P(out), M, D, D, 0, 0, 0.0, None, 0, st())),
        ("out_projection_bwd_data", 2.0 * M * D * D, lambda: lib.satrn_linear_bwd_data(dt, P(dout), D, P(wo_b), D, P(datt), M, D, D, 0, st())),
        ("out_projection_bwd_weight", 2.0 * M * D * D, lambda: lib.satrn_linear_bwd_weight(dt, P(dout), D, P(att), P(dwo), P(dbo), M, D, D, st())),
        ("attention_bwd", 10.0 * B * H * L * L * hd, lambda: lib.satrn_attention_bwd(dt, q_, k_, v_, P(att), P(lse), P(datt), dq_, dk_, dv_, P(ws), B, H, L, L, hd, 3 * D, 3 * D, 3 * D, D, 0, None, 0, 2, temp, 0.0, None, 0, st())),
        ("qkv_projection_bwd_data", 2.0 * M * 3 * D * D, lambda: lib.satrn_linear_bwd_data(dt, P(dqkv), 3 * D, P(wqkv_b), 3 * D, P(dy), M, 3 * D, D, 0, st())),
        ("qkv_projection_bwd_weight", 2.0 * M * 3 * D * D, lambda: lib.satrn_linear_bwd_weight(dt, P(dqkv), 3 * D, P(y), P(dwqkv), P(dbqkv), M, 3 * D, D, st())),
    ]
    if dt == 1 and B * L > 0:
        # round 3: the forward half of the region as ONE launch (+ the LayerNorm behind the block, which folds the partial projections)
        y2, o_full, mr2 = torch.empty(M, D, dtype=tdt, device=dev), torch.empty(M, D, dtype=tdt, device=dev), torch.empty(2 * M, device=dev)
        parts = torch.empty(H // 2, M, D, dtype=tdt, device=dev)
        fl_fwd = 2.0 * M * 3 * D * D + 4.0 * B * H * L * L * hd + 2.0 * M * D * D
        ops.insert(0, ("FUSED_region_fwd (LN + qkv + attention + out-proj partials) + LN(fold)", fl_fwd,
                       lambda: lib.satrn_enc_attn_region_fwd(P(x), P(lnw), P(lnb), P(wqkv_f), P(bqkv), P(wo_f), P(bo), B, L, D, H, 0.0, 0.0, None, 0, 0, P(y), P(mr), P(qkv),
                                                             P(att), P(lse), P(parts), P(o_full), P(y2), P(mr2), st())))
    for _, _, f in ops:  # warm-up
        ok(f())
    torch.cuda.synchronize()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(len(ops) + 1)] for _ in range(a.iters)]
    for it in range(a.iters):
        ev[it][0].record()
        for i, (_, _, f) in enumerate(ops):
            f()
            ev[it][i + 1].record()
    torch.cuda.synchronize()
    peak = 2500.0 if dt else 157.3
    res = []
    for i, (name, fl, _) in enumerate(ops):
        ts = sorted(ev[it][i].elapsed_time(ev[it][i + 1]) for it in range(a.iters))
        med = ts[len(ts) // 2] * 1e3
        res.append(dict(op=name, us_median=round(med, 2), flops=fl, tflops=round(fl / med / 1e6, 2), mfma_frac_from_flops=round(fl / med / 1e6 / peak, 4)))
    gemm_fl = sum(r["flops"] for r in res if "projection" in r["op"])
    gemm_us = sum(r["us_median"] for r in res if "projection" in r["op"])
    print(json.dumps(dict(region="encoder self-attention (LN -> QKV -> attention -> out-proj, fwd + bwd)", dtype=a.dtype, B=B, tokens=L, d=D,
                          heads=H, ops=res, projection_gemms=dict(flops=gemm_fl, us=round(gemm_us, 2), tflops=round(gemm_fl / gemm_us / 1e6, 2),
                                                                  mfma_frac_from_flops=round(gemm_fl / gemm_us / 1e6 / peak, 4)),
                          note="HIP-event timing includes the event records between the launches (~1-2 us each)")))


if __name__ == "__main__":
    main()
