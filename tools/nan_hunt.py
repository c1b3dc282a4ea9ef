import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
drop = float(sys.argv[1]); sync = int(sys.argv[2])
torch.manual_seed(21)
model = bench.make_model("bf16", 128, 384, drop).to("cuda"); model.train()
img, exp = bench.synth(32, 128, 384, 128, 21, "cuda")
out = []
for i in range(30):
    model.train_step(img, exp, 5e-4, use_graph=True)
    if sync or i % 5 == 4:
        out.append((i, round(model.read_loss()[0], 3)))
print("drop", drop, "sync", sync, out)
