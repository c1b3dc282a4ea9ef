"""CPU: the oracle's restatement of DecodingManager / MemoryNode against vectors produced by the reference's own
DecodingManager (tests/golden/rules.npz, generator: tests/golden/make_golden_rules.py), and the rule compiler."""
import os

import numpy as np
import torch

from oracle import satrn_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rules.npz")


def test_oracle_sift_reproduces_the_reference_manager_step_by_step():
    z = np.load(GOLD)
    table, logits = z["table"], torch.from_numpy(z["logits"])
    S, B, V = logits.shape
    state = O.sift_new_state(B, table)
    fired_limit = fired_balance = 0
    for t in range(S):
        mask = np.stack([O.sift_blacklist(*st, table) for st in state])
        assert (mask == z["mask"][t]).all(), f"blacklist differs at step {t}"
        cur = [st[0] for st in state]
        fired_limit += sum(bool(mask[b, cur[b]]) and cur[b] not in (0,) for b in range(B))
        fired_balance += int(mask[:, int(table[V + 5])].sum())
        tg, pr = O.sift(logits[t], state, table)
        assert (tg.numpy() == z["targets"][t]).all(), f"targets differ at step {t}"
        assert np.allclose(pr.double().numpy()[:, z["sample_pos"]], z["probs_samples"][t], rtol=0, atol=1e-7)
    # the synthetic inputs must actually exercise the run-length and the bracket-balance rules
    assert fired_limit > 10 and fired_balance > 10


def test_rule_compiler_on_a_synthetic_manager():
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    decoding = importlib.import_module("satrn_amd").decoding

    class M:
        tokens = ["<SOS>", "<EOS>", "<PAD>", "{", "}", "_", "a", "b", ""]
        rules = {"cannot_initial": ["}", "b"], "next_underbar": ["a"], "next_lbracket": [], "cannot_next_underbar": ["b"],
                 "cannot_next_lbracket": ["{"], "limit_series": {"a": False, "b": True, "{": True, "<EOS>": False},
                 "limit_params": {"b": 2, "{": 1}}
    t = decoding.compile_rules(M())
    V = len(M.tokens)
    assert t[V:V + 6].tolist() == [0, 1, 8, 5, 3, 4]
    assert t[4] == decoding.F_NOT_INITIAL and t[7] == (decoding.F_NOT_INITIAL | decoding.F_NOT_UNDERBAR | (2 << 8))
    assert t[6] == decoding.F_NEXT_UNDERBAR and t[3] == (decoding.F_NOT_LBRACKET | (1 << 8))
    # after <SOS>: <SOS>, "", "}" (balanced) and the cannot_initial tokens; after "a": everything but "_"
    assert np.flatnonzero(O.sift_blacklist(0, 1, 0, 0, t)).tolist() == [0, 4, 7, 8]
    assert np.flatnonzero(~O.sift_blacklist(6, 1, 0, 0, t)).tolist() == [5]
    assert np.flatnonzero(O.sift_blacklist(7, 2, 1, 0, t)).tolist() == [0, 5, 7, 8]  # "b" twice, one "{" open
