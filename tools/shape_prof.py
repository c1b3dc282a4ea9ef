"""Per-launch event profile of one training step split by problem size (SATRN_PROF=shapes)."""
import os, sys
os.environ["SATRN_PROF"] = "shapes"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(21)
    model = bench.make_model("bf16", 128, 384, 0.1).to(dev)
    model.train()
    img, exp = bench.synth(int(os.environ.get("B", 32)), 128, 384, 128, 21, dev)
    for _ in range(2):
        model.train_step(img, exp, 1e-3)
    torch.cuda.synchronize()
    rows = model.profile_step(img, exp)
    rows = model.profile_step(img, exp)
    tot = sum(r["ms"] for r in rows)
    print("total ms", tot)
    for r in rows[:int(os.environ.get("TOP", 70))]:
        print(f'{r["kernel"]:<60} n={r["launches"]:4d} ms={r["ms"]:.3f} us/launch={1e3*r["ms"]/r["launches"]:.1f}')

main()
