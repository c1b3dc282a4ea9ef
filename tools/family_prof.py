"""Per-family launch counts and event times of one training step (profile_step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
model = bench.make_model("bf16", 128, 384, 0.1).to(dev); model.train()
img, exp = bench.synth(32, 128, 384, 128, 21, dev)
for _ in range(2):
    model.train_step(img, exp, 1e-3)
rows = model.profile_step(img, exp); rows = model.profile_step(img, exp)
print(sum(r["launches"] for r in rows), "launches", round(sum(r["ms"] for r in rows), 3), "ms")
for r in rows:
    print(f"{r['kernel']:<28} n={r['launches']:4d} ms={r['ms']:.3f} us/launch={1e3 * r['ms'] / r['launches']:.1f}")
