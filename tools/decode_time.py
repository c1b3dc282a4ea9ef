"""Greedy decode timing, pipelined (role-per-workgroup, weights in LDS) vs one-workgroup-per-image decoder.
   python tools/decode_time.py [B] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from satrn_amd import switches as sw

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 231
dev = torch.device("cuda", 0)
torch.manual_seed(21)
model = bench.make_model("bf16", 128, 384, 0.1).to(dev)
model.eval()
img, _ = bench.synth(B, 128, 384, 4, 5, dev)
model.encode(img); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): model.encode(img)
torch.cuda.synchronize()
enc = (time.perf_counter() - t0) / 3 * 1e3
for name, env in (("per-image", "1"), ("pipelined", None)):
    if env: sw.off("decode_pipe")
    else: sw.on("decode_pipe")
    model.greedy(img, steps); torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n): model.greedy(img, steps)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"{name:10s} B={B} steps={steps}: {ms:.2f} ms per batch ({B * steps / ms * 1e3:.0f} tok/s), encoder {enc:.2f} ms, {(ms - enc) / steps * 1e3:.1f} us per step")
