import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda", 0)
H, W, T, B = 128, 384, 128, 32
torch.manual_seed(21)
model = bench.make_model("bf16", H, W, 0.1).to(dev); model.train()
img, exp = bench.synth(B, H, W, T, 5, dev)
which = sys.argv[1]
if "n" not in which:
    for _ in range(5): model.train_step(img, exp, 5e-4)
torch.cuda.synchronize()
if "p" in which:
    model.profile_step(img, exp)
if "a" in which:
    for _ in range(3): model.train_step(img, exp, 5e-4, teacher_forced=False)
    torch.cuda.synchronize()
if "d" in which:
    model.eval()
    dimg = torch.randn(64, 1, H, W, device=dev)
    model.greedy(dimg, 231); torch.cuda.synchronize()
    model.train()
if "r" in which:
    del model
    import gc; gc.collect(); torch.cuda.empty_cache()
if "e" in which:
    os.environ["SATRN_OFF"] = "side_stream"
r, f = bench.precision_report(H, W, T, B, dev)
print(which, f["ms_per_step"])
