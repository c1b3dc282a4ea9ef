"""Per-launch event profile of one SwinTRN training step split by problem size (SATRN_PROF=shapes; run on the GPU box)."""
import os, sys
os.environ["SATRN_PROF_SHAPES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import satrn_amd

dev = torch.device("cuda", 0)
flags = satrn_amd.Flags(dict(network="SWIN", input_size=dict(height=384, width=384),
                             SATRN=dict(encoder=dict(hidden_dim=300, filter_dim=600, layer_num=6, head_num=8),
                                        decoder=dict(src_dim=1024, hidden_dim=512, filter_dim=512, layer_num=4, head_num=8)),
                             data=dict(rgb=3), dropout_rate=0.1)).get()
torch.manual_seed(21)
m = satrn_amd.SWIN(flags, bench._DS(), True, dtype="bf16").to(dev)
m.train()
g = torch.Generator().manual_seed(5)
B, T = 16, 128
img = torch.randn(B, 3, 384, 384, generator=g).to(dev)
exp = torch.randint(3, 245, (B, T + 1), generator=g)
exp[:, 0] = 0
exp[:, -1] = 1
exp = exp.to(dev)
for _ in range(3):
    m.train_step(img, exp, 5e-4)
torch.cuda.synchronize()
rows = m.profile_step(img, exp)
rows = m.profile_step(img, exp)
print("total ms", sum(r["ms"] for r in rows))
for r in rows[:int(os.environ.get("TOP", 60))]:
    us = 1e3 * r["ms"] / r["launches"]
    fl = r["flops"] / r["launches"]
    by = r["bytes"] / r["launches"]
    print(f'{r["kernel"]:<58} n={r["launches"]:4d} ms={r["ms"]:7.3f} us/launch={us:7.1f}  {fl / us / 1e6:7.1f} TF  {by / us / 1e3:7.0f} GB/s')
