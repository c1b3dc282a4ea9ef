"""Phase marks of the MBConv front kernel inside the training step (GPU box):  SATRN_PROF=mb python tools/mbconv_prof.py 2> marks.txt
Every launch of the kernel synchronises and prints the wall-clock marks of its four corner workgroups (kernels_mbconv.hip)."""
import os, sys
os.environ.setdefault("SATRN_PROF", "mb")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
m = bench.make_model("bf16", 128, 384, 0.1).to(dev)
m.train()
img, exp = bench.synth(32, 128, 384, 128, 21, dev)
for _ in range(3):
    print("---- step", file=sys.stderr)
    m.train_step(img, exp, 5e-4)
    torch.cuda.synchronize()
