"""Time the train-time autoregressive branch (teacher_forcing_ratio = 0: networks/EfficientSATRN.py:496-525) forward + backward
next to the teacher-forced one, EfficientSATRN bs32 128x384 T=128 bf16."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

model = bench.make_model("bf16", 128, 384, 0.1).to("cuda")
model.train()
img, exp = bench.synth(int(os.environ.get("B", 32)), 128, 384, int(os.environ.get("T", 128)), 21, "cuda")
for tf in (1.0, 0.0):
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        logits = model(img, exp, True, tf)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        loss = model.criterion(logits.transpose(1, 2), exp[:, 1:])
        model.zero_grad()
        loss.backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"tf={tf}: forward {1e3*(t1-t0):.1f} ms, loss+backward {1e3*(t2-t1):.1f} ms, loss {loss.item():.4f}", flush=True)
