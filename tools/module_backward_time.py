import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda:0")
H, W, T, B = 128, 384, 128, 32
torch.manual_seed(21)
model = bench.make_model("bf16", H, W, 0.1).to(dev)
model.train()
img, exp = bench.synth(B, H, W, T, 21, dev)
for mode in ("autograd", "direct", "autograd", "direct"):
    ts = []
    for i in range(4):
        logits = model(img, exp, True, 1.0)
        loss = model.criterion(logits.transpose(1, 2), exp[:, 1:])
        if mode == "direct":
            dl = torch.autograd.grad(loss, logits, retain_graph=False)[0] if False else torch.ones_like(logits) / logits.numel()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == "autograd":
            model.zero_grad(); loss.backward()
        else:
            model.zero_grad(); model._run_backward(dl)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    print(mode, ["issue %.2f total %.2f" % t for t in ts])
