"""Per-step metrics: oracle vs the reference's own id_to_string / sentence_acc / symbol counts
(tests/golden/metrics.npz; the edit distance itself is pinned to the Levenshtein definition, the reference's `editdistance`
package is absent), then the device kernel vs the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import satrn_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "metrics.npz")


@pytest.mark.parametrize("case", ["a", "b"])
def test_oracle_metrics_match_reference_strings(case):
    z = np.load(GOLD)
    seq, exp = torch.from_numpy(z[case + "_sequence"]), torch.from_numpy(z[case + "_expected"])
    B = seq.shape[0]
    npred = [len(O.metric_tokens(seq[b].tolist())) for b in range(B)]
    ngt = [len(O.metric_tokens(exp[b].tolist())) for b in range(B)]
    assert npred == z[case + "_ntok_pred"].tolist() and ngt == z[case + "_ntok_gt"].tolist()
    eq = [O.metric_tokens(seq[b].tolist()) == O.metric_tokens(exp[b].tolist()) for b in range(B)]
    assert eq == z[case + "_equal"].tolist()
    m = O.step_metrics(seq, exp)
    assert m["correct_sentences"] / B == pytest.approx(float(z[case + "_sent_acc"]))
    assert m["correct_symbols"] == int(z[case + "_correct_symbols"]) and m["total_symbols"] == int(z[case + "_total_symbols"])
    assert m["sum_wer"] == pytest.approx(float(z[case + "_sum_wer_oracle"]))


def test_levenshtein_known_answers():
    assert O.levenshtein(list("kitten"), list("sitting")) == 3
    assert O.levenshtein([], [1, 2, 3]) == 3 and O.levenshtein([1, 2, 3], [1, 2, 3]) == 0
    assert O.levenshtein([1, 2, 3, 4], [2, 3, 4, 5]) == 2


@pytest.mark.gpu
def test_device_step_metrics_match_oracle():
    import satrn_amd
    z = np.load(GOLD)
    t2i = {"<PAD>": 2, "<SOS>": 0, "<EOS>": 1, "": O.NUM_CLASSES - 1}
    sm = satrn_amd.StepMetrics(t2i)
    tot = dict(sum_wer=0.0, sentences=0, correct_sentences=0, correct_symbols=0, total_symbols=0)
    for case in ("a", "b"):
        seq, exp = torch.from_numpy(z[case + "_sequence"]), torch.from_numpy(z[case + "_expected"])
        exp_m1 = exp.clone()
        exp_m1[exp_m1 == 2] = -1  # the trainer's in-place PAD -> -1 (train_single_opt.py:101) must give the same numbers
        sm.update(seq.cuda(), (exp if case == "a" else exp_m1).cuda())
        m = O.step_metrics(seq, exp)
        for k in tot:
            tot[k] += m[k]
    r = sm.result()
    assert r["sentences"] == tot["sentences"] and r["correct_sentences"] == tot["correct_sentences"]
    assert r["correct_symbols"] == tot["correct_symbols"] and r["total_symbols"] == tot["total_symbols"]
    assert r["sum_wer"] == pytest.approx(tot["sum_wer"], rel=1e-12)
    # long random sequences (anti-diagonal DP across several 64-lane strips), ragged lengths, strided views
    g = torch.Generator().manual_seed(5)
    B, T = 9, 300
    exp = torch.randint(3, 245, (B, T + 1), generator=g)
    exp[:, 0] = 0
    seq = exp[:, 1:].clone()
    for b in range(B):
        n = 5 + 31 * b
        exp[b, n] = 1
        exp[b, n + 1:] = 2
        flip = torch.rand(T, generator=g) < 0.15 * (b % 3)
        seq[b][flip] = torch.randint(0, 245, (int(flip.sum()),), generator=g)
        seq[b, min(T - 1, n - 1 + (b % 4) - 1)] = 1
    sm.reset()
    big = torch.zeros(B, T + 7, dtype=torch.int64)
    big[:, :T] = seq
    sm.update(big.cuda()[:, :T], exp.cuda())
    r, m = sm.result(), O.step_metrics(seq, exp)
    assert (r["sentences"], r["correct_sentences"], r["correct_symbols"], r["total_symbols"]) == (
        m["sentences"], m["correct_sentences"], m["correct_symbols"], m["total_symbols"])
    assert r["sum_wer"] == pytest.approx(m["sum_wer"], rel=1e-12)


def test_id_to_string_follows_the_reference_loop():
    """utils/utils.py:134-164 restated literally (per-token loop) against the one-transfer version."""
    import satrn_amd

    class DS:
        token_to_id = {"<SOS>": 0, "<EOS>": 1, "<PAD>": 2}
        id_to_token = {i: f"t{i}" for i in range(245)}
    DS.id_to_token[244] = ""

    class DL:
        dataset = DS()

    def ref(tokens, do_eval):
        out = []
        special = {2, 0, 1}
        for ex in tokens:
            s = ""
            for tok in ex:
                tok = tok.item()
                if do_eval:
                    if tok not in special:
                        if tok != -1:
                            s += DS.id_to_token[tok] + " "
                    elif tok == 1:
                        break
                elif tok != -1:
                    s += DS.id_to_token[tok] + " "
            out.append(s)
        return out

    g = torch.Generator().manual_seed(3)
    toks = torch.randint(-1, 245, (6, 17), generator=g)
    toks[0, 5] = 1; toks[1, 0] = 1; toks[2, :] = 2; toks[3, 3] = 0; toks[4, 7] = 244
    for do_eval in (0, 1):
        assert satrn_amd.id_to_string(toks, DL, do_eval) == ref(toks, do_eval)
    assert satrn_amd.id_to_string(toks[:0], DL, 1) == []
