"""Robustness sweep: bf16 training steps + greedy decode of EfficientSATRN at several batch sizes / resolutions / lengths;
every loss must be finite and fall over a few steps."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda", 0)
for (B, H, W, T) in [(1, 128, 384, 16), (7, 128, 384, 40), (64, 128, 384, 128), (5, 64, 256, 9), (3, 96, 160, 31), (2, 160, 512, 64)]:
    torch.manual_seed(B)
    model = bench.make_model("bf16", H, W, 0.1).to(dev)
    model.train()
    img, exp = bench.synth(B, H, W, T, 3, dev)
    losses = []
    for i in range(6):
        model.train_step(img, exp, 1e-3)
        losses.append(model.read_loss()[0])
    model.eval()
    lg, ids = model.greedy(img, min(T, 12))
    ok = all(math.isfinite(x) for x in losses) and losses[-1] < losses[0] and bool(torch.isfinite(lg).all())
    print(f"B={B} {H}x{W} T={T}: loss {losses[0]:.3f} -> {losses[-1]:.3f}  greedy ok={bool(torch.isfinite(lg).all())}  {'OK' if ok else 'FAIL'}", flush=True)
    assert ok
    del model
    torch.cuda.empty_cache()
print("sweep ok")
