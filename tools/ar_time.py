"""Train-time autoregressive branch (teacher_forcing_ratio = 0: networks/EfficientSATRN.py:496-525) forward + backward, module
API, EfficientSATRN bs32 128x384 T=128, next to the teacher-forced branch (run on the GPU box).
SATRN_TIMING_SKIP_WGRAD=1 drops the weight-gradient launches (wrong gradients): what the dependent chain alone costs."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda:0")
H, W, T, B = 128, 384, 128, 32
torch.manual_seed(21)
model = bench.make_model("bf16", H, W, 0.1).to(dev)
model.train()
img, exp = bench.synth(B, H, W, T, 21, dev)


def run(tf_ratio, n):
    fw = bw = 0.0
    for i in range(n + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        logits = model(img, exp, True, tf_ratio)
        loss = model.criterion(logits.transpose(1, 2), exp[:, 1:])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        model.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if i:
            fw += t1 - t0
            bw += t2 - t1
    return fw / n * 1e3, bw / n * 1e3


for name, r in (("teacher-forced", 1.0), ("autoregressive", 0.0)):
    run(r, 3)  # warm-up: the autograd thread's first backward calls take 50-70 ms
    f, b = run(r, 4)
    print(f"{name:15s}: forward {f:7.2f} ms, backward {b:7.2f} ms, total {f + b:7.2f} ms")

# the fused step (one library call: forward + CE + backward + clip + AdamW), both branches
for name, tf in (("fused teacher-forced", True), ("fused autoregressive", False)):
    for _ in range(3):
        model.train_step(img, exp, 5e-4, teacher_forced=tf)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        model.train_step(img, exp, 5e-4, teacher_forced=tf)
    torch.cuda.synchronize()
    print(f"{name:22s}: {(time.perf_counter() - t0) / n * 1e3:7.2f} ms/step")
