"""Golden vectors for the DecodingManager path, produced by running THE REFERENCE's own DecodingManager
(postprocessing/postprocessing.py, with its RULES and configs/tokens.txt) on CPU in the authoring container.

    python tests/golden/make_golden_rules.py

Stored in tests/golden/rules.npz (data only):
  table         int32 [V+8]   the reference's RULES compiled by satrn_amd.decoding.compile_rules (flags | limit << 8, ids)
  logits        f32 [S,B,V]   synthetic step inputs (deterministic; biased so run-length and bracket rules fire)
  targets       i64 [S,B]     what DecodingManager.sift returned, step by step
  mask          bool [S,B,V]  the blacklist the reference applied at each step
  probs_samples f64 [S,B,8]   masked probabilities at 8 fixed vocabulary positions
  lite_*                      LiteSATRN (deterministic weights) greedy decode WITH the manager: ids, probability samples
The reference calls manager.reset() without its required argument at the end of a managed decode
(networks/LiteSATRN.py:542-543), which raises; the harness gives that one parameter a default (reference file untouched).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (reference import harness)
from oracle import satrn_oracle as O  # noqa: E402


def synth_logits(S, B, V, table):
    lbr, rbr = int(table[V + 4]), int(table[V + 5])
    x = O.det_tensor((S, B, V), 777, 3.0).clone()
    h = O.det_tensor((S, B, 4), 778, 1.0)
    return x, h, lbr, rbr


def main():
    utils, LiteSATRN, EfficientSATRN = G.import_reference()
    sys.path.insert(0, os.path.join(G.ROOT, "p4-fr-sorry-math-but-love-you_amd"))
    import importlib.util
    spec = importlib.util.spec_from_file_location("satrn_amd_pkg", os.path.join(G.ROOT, "p4-fr-sorry-math-but-love-you_amd", "__init__.py"),
                                                  submodule_search_locations=[os.path.join(G.ROOT, "p4-fr-sorry-math-but-love-you_amd")])
    pkg = importlib.util.module_from_spec(spec)
    sys.modules["satrn_amd_pkg"] = pkg
    spec.loader.exec_module(pkg)
    from satrn_amd_pkg.decoding import compile_rules
    from postprocessing.postprocessing import get_decoding_manager, DecodingManager

    _reset = DecodingManager.reset
    DecodingManager.reset = lambda self, sequence_length=None: _reset(self, sequence_length)

    S, B = 48, 6
    manager = get_decoding_manager(os.path.join(G.REF, "configs/tokens.txt"), batch_size=B)
    V = manager.vocab_size
    table = compile_rules(manager)
    x, h, lbr, rbr = synth_logits(S, B, V, table)
    manager.reset(sequence_length=S)
    targets, masks, samples = [], [], []
    prev = torch.full((B,), int(table[V]), dtype=torch.int64)
    pos = (np.arange(8) * 31 + 3) % V
    for t in range(S):
        step = x[t].clone()
        for b in range(B):  # push towards repeats / brackets so the run-length and balance rules are exercised
            if h[t, b, 0] > 0.0:
                step[b, prev[b]] += 9.0
            if h[t, b, 1] > 0.55:
                step[b, lbr] += 8.0
            if h[t, b, 2] > 0.35:
                step[b, rbr] += 8.5
            if h[t, b, 3] > 0.85:
                step[b, 1] += 12.0  # <EOS>
        x[t] = step
        masks.append(np.stack([DecodingManager._mask(n, V).numpy() for n in manager.memories]))
        tg, pr = manager.sift(step)
        targets.append(tg.numpy().astype(np.int64))
        samples.append(pr.double().numpy()[:, pos])
        prev = tg
    out = dict(table=table, logits=x.numpy().astype(np.float32), targets=np.stack(targets), mask=np.stack(masks),
               probs_samples=np.stack(samples), sample_pos=pos.astype(np.int64))
    hits = np.stack(masks).sum(-1)
    print("blacklist sizes: min", hits.min(), "max", hits.max(), "; distinct targets", len(set(np.stack(targets).flatten().tolist())))

    # ---- model level: LiteSATRN greedy decode with the manager (eval mode, deterministic weights)
    cfg = dict(O.CFG_LITE)
    mb = 3
    manager2 = get_decoding_manager(os.path.join(G.REF, "configs/tokens.txt"), batch_size=mb)
    ds = G._DS()
    ds.token_to_id, ds.id_to_token = utils.load_vocab([os.path.join(G.REF, "configs/tokens.txt")])
    model = LiteSATRN(G.flags_for(utils, cfg, 64, 192), ds, None, manager2)
    sd = O.det_state_dict(cfg, 9)
    model.load_state_dict(sd, strict=True)
    model.eval()
    img, expected = O.det_inputs(mb, 1, 64, 192, 12, seed=70)
    with torch.no_grad():
        probs = model(img, expected, False, 0.0)  # [b, steps, V] masked softmax probabilities
    top2 = torch.topk(probs, 2, dim=-1)
    out["lite_ids"] = top2.indices[..., 0].numpy().astype(np.int64)
    out["lite_margin"] = (top2.values[..., 0] - top2.values[..., 1]).numpy()
    out["lite_probs_samples"] = probs.double().numpy()[..., pos]
    out["lite_meta"] = np.array([mb, 64, 192, 12, 9, 70], dtype=np.int64)  # batch, H, W, steps, wseed, iseed
    path = os.path.join(HERE, "rules.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; lite ids", out["lite_ids"][0].tolist())


if __name__ == "__main__":
    main()
