import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda:0")
H, W, T, B = 128, 384, 128, 32
torch.manual_seed(21)
model = bench.make_model("bf16", H, W, 0.1).to(dev)
model.train()
img, exp = bench.synth(B, H, W, T, 21, dev)
for i in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    logits = model(img, exp, True, 0.0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    loss = model.criterion(logits.transpose(1, 2), exp[:, 1:])
    model.zero_grad()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    loss.backward()
    t4 = time.perf_counter()
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    if i >= 4:
        print(f"forward: issue {1e3*(t1-t0):.1f} ms, done {1e3*(t2-t0):.1f} ms | backward: issue {1e3*(t4-t3):.1f} ms, done {1e3*(t5-t3):.1f} ms")
