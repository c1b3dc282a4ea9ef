"""64 x 231 greedy decode with the DecodingManager rules on the device (the reference's default at inference), pipelined vs
per-image decoder (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import satrn_amd
from satrn_amd import switches as sw

dev = torch.device("cuda", 0)
torch.manual_seed(21)
model = bench.make_model("bf16", 128, 384, 0.1).to(dev)
model.eval()
table = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "rules.npz"))["table"]


class _M:
    tokens = ["<SOS>", "<EOS>"] + [f"t{i}" for i in range(len(table) - 10)]
    rules = {}


mgr = satrn_amd.DeviceDecodingManager(_M())
mgr._table_host = table.astype(np.int32)
model.decoder.manager = mgr
img, _ = bench.synth(64, 128, 384, 4, 5, dev)
for name, env in (("per-image", "1"), ("pipelined", None)):
    if env: sw.off("decode_pipe")
    else: sw.on("decode_pipe")
    model.greedy(img, 231); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): model.greedy(img, 231)
    torch.cuda.synchronize()
    print(f"{name:10s} with rules: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per 64 x 231 batch")
