"""Time the stride-2 3x3 data gradient (C-ABI satrn_conv3x3_bwd_data) on the two stage entries of EfficientNetV2-S at the benchmark
batch, parity-class form against the all-taps form (SATRN_OFF=dgrad_classes); tile kernel (SATRN_KNOBS=conv_big=0) and default routing."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    return a.elapsed_time(b) / iters * 1e3  # us

for B, H, W, Ci, Co in [(32, 64, 192, 24, 96), (32, 32, 96, 48, 192)]:
    OH, OW = H // 2, W // 2
    w = (torch.randn(Co, Ci, 3, 3, device="cuda") * 0.05).bfloat16()
    fwd = torch.empty(Co, 9, Ci, dtype=torch.bfloat16, device="cuda"); bwd = torch.empty(Ci, 9, Co, dtype=torch.bfloat16, device="cuda")
    assert lib.satrn_pack_conv3x3(1, P(w), P(fwd), P(bwd), Co, Ci, st()) == 0
    dy = torch.randn(B, OH, OW, Co, device="cuda").bfloat16()
    dx = torch.empty(B, H, W, Ci, dtype=torch.bfloat16, device="cuda")
    for label, env in (("classes, tile kernel (default)", {}), ("all taps, tile kernel", {"SATRN_OFF": "dgrad_classes", "SATRN_KNOBS": "conv_big=0"}),
                       ("persistent kernel where it applies", {"SATRN_OFF": "dgrad_classes_tile"}),
                       ("classes, 256-row tiles (<= 32 ch)", {"SATRN_KNOBS": "dgrad_bm=256"}), ("classes, 64-row tiles (<= 32 ch)", {"SATRN_KNOBS": "dgrad_bm=64"})):
        for k in ("SATRN_OFF", "SATRN_KNOBS"): os.environ.pop(k, None)
        os.environ.update(env)
        us = bench(lambda: lib.satrn_conv3x3_bwd_data(1, P(dy), P(bwd), P(dx), B, H, W, Ci, Co, OH, OW, 2, 0, 0, 0, st()))
        mb = (dy.numel() + dx.numel()) * 2 / 1e6
        print(f"dgrad s2 {Co:3d} -> {Ci:2d} ch, {B}x{H}x{W}: {label:36s} {us:7.1f} us  ({mb / us:.2f} TB/s algorithmic)")
