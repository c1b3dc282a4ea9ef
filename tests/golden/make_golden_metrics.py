"""Golden vectors for the per-step metrics: the reference's own id_to_string (utils/utils.py:134-164) and sentence_acc
(utils/metrics.py:26-34) on synthetic predictions, plus the symbol counts of train_modules/train_single_opt.py:108-109.
word_error_rate needs the third-party `editdistance` package (absent here), so the WER column is NOT reference output.

    python tests/golden/make_golden_metrics.py   ->  tests/golden/metrics.npz
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402
from oracle import satrn_oracle as O  # noqa: E402


def synth(B, T, seed):
    """ground truth rows <SOS> tokens <EOS> <PAD>...; predictions = ground truth with substitutions, early / late / missing
    <EOS>, every 4th row perfect; the arrays themselves are stored in the fixture"""
    V = O.NUM_CLASSES
    exp = (O.det_tensor((B, T + 1), seed, 1.0).abs() * 1e4).long() % (V - 3) + 3
    exp[:, 0] = O.SOS_ID
    seq = exp[:, 1:].clone()
    noise = O.det_tensor((B, T), seed + 1, 1.0)
    repl = (O.det_tensor((B, T), seed + 2, 1.0).abs() * 1e4).long() % V
    for b in range(B):
        n = 3 + (7 * b) % (T - 3)              # position of <EOS> in the ground truth
        exp[b, n] = 1
        exp[b, n + 1:] = O.PAD_ID
        seq[b, n - 1] = 1                      # predicted <EOS> where the ground truth has it
        if b % 4 != 0:                         # every 4th prediction is perfect
            m = noise[b].abs() > 0.7
            seq[b][m] = repl[b][m]
        if b % 5 == 1:
            seq[b, min(T - 1, n + 1)] = 1      # an extra, later <EOS>
    seq[1, 2] = V - 1                          # the "" token in a prediction
    return seq, exp


def main():
    utils, _, _ = G.import_reference()
    from utils.metrics import sentence_acc

    class _DL:
        class dataset:
            pass
    _DL.dataset.token_to_id, _DL.dataset.id_to_token = utils.load_vocab([os.path.join(G.REF, "configs/tokens.txt")])
    out = {}
    for name, (B, T, seed) in dict(a=(12, 20, 300), b=(5, 7, 310)).items():
        seq, exp = synth(B, T, seed)
        e2 = exp.clone()
        e2[e2 == O.PAD_ID] = -1                                                   # train_single_opt.py:101
        es = utils.id_to_string(e2, _DL, do_eval=1)
        ss = utils.id_to_string(seq, _DL, do_eval=1)
        out[name + "_meta"] = np.array([B, T, seed], dtype=np.int64)
        out[name + "_sequence"] = seq.numpy().astype(np.int64)
        out[name + "_expected"] = exp.numpy().astype(np.int64)
        out[name + "_ntok_pred"] = np.array([len(x.split(" ")) for x in ss], dtype=np.int64)
        out[name + "_ntok_gt"] = np.array([len(x.split(" ")) for x in es], dtype=np.int64)
        out[name + "_equal"] = np.array([a == b for a, b in zip(ss, es)])
        out[name + "_sent_acc"] = np.array(sentence_acc(ss, es), dtype=np.float64)
        out[name + "_correct_symbols"] = np.array(torch.sum(seq == e2[:, 1:], dim=(0, 1)).item(), dtype=np.int64)
        out[name + "_total_symbols"] = np.array(torch.sum(e2[:, 1:] != -1, dim=(0, 1)).item(), dtype=np.int64)
        m = O.step_metrics(seq, exp)
        out[name + "_sum_wer_oracle"] = np.array(m["sum_wer"], dtype=np.float64)     # NOT reference output (editdistance absent)
        print(name, "sent_acc", out[name + "_sent_acc"], "oracle", m)
        assert m["correct_sentences"] == int(out[name + "_equal"].sum())
    np.savez_compressed(os.path.join(HERE, "metrics.npz"), **out)


if __name__ == "__main__":
    main()
