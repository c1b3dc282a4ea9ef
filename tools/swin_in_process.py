"""SwinTRN step measured alone and behind other models of the same process (stream / hardware-queue aliasing check)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
which = sys.argv[1] if len(sys.argv) > 1 else "x"
H, W, T, B = 128, 384, 128, 32
if "m" in which:
    torch.manual_seed(21)
    model = bench.make_model("bf16", H, W, 0.1).to(dev); model.train()
    img, exp = bench.synth(B, H, W, T, 5, dev)
    for _ in range(5): model.train_step(img, exp, 5e-4)
    torch.cuda.synchronize()
if "f" in which:
    bench.precision_report(H, W, T, B, dev)
r = bench.swin_report(dev)
print(which, r["ms_per_step"])
