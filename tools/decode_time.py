import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
model = bench.make_model("bf16", 128, 384, 0.1).to("cuda"); model.eval()
NB = int(os.environ.get("NB", 64))
img, _ = bench.synth(NB, 128, 384, 4, 5, "cuda")
model.greedy(img, 231); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(3): model.greedy(img, 231)
torch.cuda.synchronize()
print("dbg", os.environ.get("SATRN_DEC_DBG", "0"), "B", NB, "ms per decode", (time.perf_counter() - t) / 3 * 1e3)
