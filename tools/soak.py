"""Soak: a few hundred bf16 training steps of the bench workload on fixed synthetic data -- the loss must fall
monotonically-ish and stay finite (overfitting one batch), gradient norm finite."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda", 0)
torch.manual_seed(21)
model = bench.make_model("bf16", 128, 384, 0.1).to(dev)
model.train()
img, exp = bench.synth(32, 128, 384, 128, 21, dev)
N = int(os.environ.get("STEPS", 300))
hist = []
for i in range(N):
    model.train_step(img, exp, 5e-4)
    if i % 25 == 0 or i == N - 1:
        loss, cnt, gn = model.read_loss()
        hist.append(loss)
        print(f"step {i:4d} loss {loss:.4f} gnorm {gn:.4f}", flush=True)
        assert math.isfinite(loss) and math.isfinite(gn)
assert hist[-1] < 0.5 * hist[0], (hist[0], hist[-1])
print("soak ok", hist[0], "->", hist[-1])
