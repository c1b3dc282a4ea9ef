"""Autoregressive training branch (train_step(teacher_forced=False)): eager two-stream launches vs one hipGraph replay.
   python tools/ar_time.py [B] [T]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda", 0)
torch.manual_seed(21)
model = bench.make_model("bf16", 128, 384, 0.1).to(dev)
model.train()
img, exp = bench.synth(B, 128, 384, T, 5, dev)
lr = 5e-4
for name, kw in (("tf eager", dict()), ("ar eager", dict(teacher_forced=False)), ("ar graph", dict(teacher_forced=False, use_graph=True))):
    for _ in range(3): model.train_step(img, exp, lr, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n): model.train_step(img, exp, lr, **kw)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"{name}: {ms:.2f} ms per step, loss {model.read_loss()[0]:.4f}", flush=True)
