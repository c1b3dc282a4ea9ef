"""Host time needed to ISSUE one training step vs the GPU time it takes (run on the GPU box).
  python3 tools/host_issue.py            # steps queued back to back
  python3 tools/host_issue.py --sync     # the GPU is idle when every step starts
With SATRN_PROF=host the engine also prints the forward / backward / optimizer split of the issue time."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
import satrn_amd  # noqa: E402


def main():
    sync = "--sync" in sys.argv
    dev = torch.device("cuda:0")
    H, W, T, B = 128, 384, 128, 32
    torch.manual_seed(21)
    model = bench.make_model("bf16", H, W, 0.1).to(dev)
    model.train()
    img, exp = bench.synth(B, H, W, T, 1, dev)
    for _ in range(4):
        model.train_step(img, exp, 1e-4)
    torch.cuda.synchronize()
    hs, gs = [], []
    for _ in range(10):
        t0 = time.perf_counter()
        model.train_step(img, exp, 1e-4)
        t1 = time.perf_counter()
        if sync:
            torch.cuda.synchronize()
            gs.append((time.perf_counter() - t0) * 1e3)
        hs.append((t1 - t0) * 1e3)
    torch.cuda.synchronize()
    print("host issue ms/step:", [round(h, 2) for h in hs])
    if sync:
        print("issue + drain ms/step:", [round(g, 2) for g in gs])


if __name__ == "__main__":
    main()
