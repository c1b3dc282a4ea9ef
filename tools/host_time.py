import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
model = bench.make_model("bf16", 128, 384, 0.1).to("cuda"); model.train()
img, exp = bench.synth(32, 128, 384, 128, 21, "cuda")
for g in (False, True):
    for _ in range(4): model.train_step(img, exp, 5e-4, use_graph=g)
    torch.cuda.synchronize()
    hs = []
    t0 = time.perf_counter()
    for _ in range(10):
        a = time.perf_counter(); model.train_step(img, exp, 5e-4, use_graph=g); hs.append(time.perf_counter() - a)
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / 10
    print(f"graph={g}: host call {1e3*sum(hs)/len(hs):.2f} ms, wall per step {1e3*tot:.2f} ms")
