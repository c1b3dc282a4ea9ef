"""Time the one-launch beam search (EfficientSATRN, 128x384, bf16) next to the greedy decode of the same batch."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import satrn_amd
from bench import make_model  # noqa


class _L:
    class dataset:
        token_to_id = {"<SOS>": 0, "<EOS>": 1, "<PAD>": 2}


def main():
    model = make_model("bf16", 128, 384, 0.1)
    model.eval()
    for B in (1, 64, 256):
        img = torch.randn(B, 1, 128, 384, device="cuda")
        for bw in (1, 5):
            for _ in range(2):
                model.beam_search(img, _L, beam_width=bw, max_sequence=230)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3):
                seq = model.beam_search(img, _L, beam_width=bw, max_sequence=230)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
            ln = (seq != 2).sum(1).float().mean().item()
            print(f"beam B={B} bw={bw}: {dt*1e3:.1f} ms/batch  mean utterance length {ln:.1f}", flush=True)
        for _ in range(2):
            model.greedy(img, 231)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            model.greedy(img, 231)
        torch.cuda.synchronize(); print(f"greedy B={B}: {(time.perf_counter()-t0)/3*1e3:.1f} ms/batch", flush=True)


if __name__ == "__main__":
    main()
