"""Golden vectors for beam search: the reference's own EfficientSATRN.beam_search (networks/EfficientSATRN.py:708-867)
run on CPU in the authoring container with build-owned deterministic weights.  The reference's LiteSATRN class has no
beam_search; the "lite" cases call the same reference function on the reference's LiteSATRN modules (it only touches
self.encoder / self.decoder), which keeps the fixtures fast to generate and small.

    python tests/golden/make_golden_beam.py   ->  tests/golden/beam.npz  (inputs are regenerated from seeds; data only)

Cases: flat distributions (breadth-first behaviour, result = best node left in the queue), <EOS> first, a sharpened
generator (the search runs deep along the likely path) and <EOS> in mid-sequence (see CASES).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402
from oracle import satrn_oracle as O  # noqa: E402

from beam_cases import CASES, weights  # noqa: E402


class _Loader:
    def __init__(self, ds):
        self.dataset = ds


def main():
    utils, LiteSATRN, EfficientSATRN = G.import_reference()
    torch.set_num_threads(8)
    out = {}
    for name, (net, cfg, B, H, W, seed, bw, ms, gs, el, eb) in CASES.items():
        cls = LiteSATRN if net == "lite" else EfficientSATRN
        model, _ = G.build_reference(utils, cls, cfg, H, W, seed)
        model.load_state_dict(weights(cfg, seed, gs, el, eb), strict=True)
        model.eval()
        img, _ = O.det_inputs(B, cfg["rgb"], H, W, 4, seed=21 + seed)
        ds = G._DS()
        ds.token_to_id, ds.id_to_token = utils.load_vocab([os.path.join(G.REF, "configs/tokens.txt")])
        with torch.no_grad():
            seq = EfficientSATRN.beam_search(model, img, _Loader(ds), topk=1, beam_width=bw, max_sequence=ms)
        out[name] = seq.numpy().astype(np.int64)
        print(name, seq.tolist())
    np.savez_compressed(os.path.join(HERE, "beam.npz"), **out)


if __name__ == "__main__":
    main()
