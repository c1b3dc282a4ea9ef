"""Cost of cutting the backward into the four exchange segments (single GPU, no collective): ms/step of phase 3 vs
phases 16..19 + 2."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda", 0)
torch.manual_seed(21)
model = bench.make_model("bf16", 128, 384, 0.1).to(dev)
model.train()
img, exp = bench.synth(32, 128, 384, 128, 21, dev)

def whole():
    model.train_step(img, exp, 5e-4)

def segmented():
    model.train_step(img, exp, 5e-4, phase=16 + 0 + 4 * 2)
    model.train_step(img, exp, 5e-4, phase=16 + 3)
    model.train_step(img, exp, 5e-4, phase=2)

for name, fn in (("whole", whole), ("segmented", segmented), ("whole", whole), ("segmented", segmented)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t) / 20 * 1e3:.3f} ms/step")
