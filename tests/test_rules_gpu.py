"""GPU: DecodingManager on the device -- the stand-alone sift entry point, the managed greedy decode
(rules inside the persistent decode kernel and on the step-wise path) and the ensemble driver loop with a manager, against
the reference manager's own vectors (tests/golden/rules.npz) and the oracle."""
import os
from satrn_amd import switches as sw

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import satrn_oracle as O
from tests.test_model_gpu import build, relerr

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rules.npz")


class _Compiled:
    """stands in for a reference DecodingManager whose rules are already compiled (the golden file carries the table)"""

    def __init__(self, table):
        self.table_host = table


def _manager(table):
    import satrn_amd
    V = len(table) - 8

    class M:  # minimal duck-typed manager; the table is injected afterwards
        tokens = ["<SOS>", "<EOS>"] + [f"t{i}" for i in range(V - 2)]
        rules = {}
        batch_size = 0
    m = satrn_amd.DeviceDecodingManager(M())
    m._table_host = np.asarray(table, dtype=np.int32).copy()
    return m


def test_device_sift_replays_the_reference_manager():
    z = np.load(GOLD)
    table, logits = z["table"], torch.from_numpy(z["logits"]).cuda()
    S, B, V = logits.shape
    m = _manager(table)
    m.reset(sequence_length=S)
    for t in range(S):
        tg, pr = m.sift(logits[t])
        assert (tg.cpu().numpy() == z["targets"][t]).all(), f"targets differ at step {t}"
        zero = pr.cpu().numpy() == 0
        assert (zero == z["mask"][t]).all(), f"mask differs at step {t}"
        assert np.allclose(pr.double().cpu().numpy()[:, z["sample_pos"]], z["probs_samples"][t], rtol=0, atol=2e-7)
    # 3-D input ([B, 1, V], what the decoder hands over) and a changed batch size restart the memories
    tg, pr = m.sift(logits[0][:2].unsqueeze(1))
    assert pr.shape == (2, 1, V) and (tg.cpu().numpy() == z["targets"][0][:2]).all()


@pytest.mark.parametrize("stepwise", [False, True])
def test_managed_greedy_decode_matches_reference_golden_and_oracle(monkeypatch, stepwise):
    z = np.load(GOLD)
    table = z["table"]
    mb, H, W, steps, wseed, iseed = (int(v) for v in z["lite_meta"])
    if stepwise:
        sw.off("decode_kernel")
    cfg = dict(O.CFG_LITE)
    model, sd = build(cfg, H, W, "f32", wseed)
    model.decoder.manager = _manager(table)
    model.eval()
    img, expected = O.det_inputs(mb, 1, H, W, steps, seed=iseed)
    probs = model(img.cuda(), expected.cuda(), False, 0.0)
    assert probs.shape == (mb, steps, cfg["num_classes"])
    ids = probs.argmax(-1).cpu().numpy()
    sure = z["lite_margin"] > 1e-4
    assert (ids[sure] == z["lite_ids"][sure]).all()
    assert np.allclose(probs.double().cpu().numpy()[..., z["sample_pos"]], z["lite_probs_samples"], rtol=0, atol=2e-5)
    # and the oracle's managed decode end to end
    oprob, oids = O.decoder_greedy_managed(O.encoder_forward(img, sd, cfg, False), steps, sd, cfg, table)
    assert (oids.numpy()[sure] == z["lite_ids"][sure]).all()
    assert relerr(probs, oprob) < 1e-4


def test_ensemble_driver_loop_with_manager():
    """utils/ensemble_utils.py:79-99 with manager is not None: manager.sift on the averaged probabilities."""
    from tests.test_ensemble_gpu import _halves
    z = np.load(GOLD)
    table = z["table"]
    cfg = dict(O.CFG_EFF)
    pairs = [_halves(cfg, 5, "f32"), _halves(cfg, 6, "f32")]
    img, _ = O.det_inputs(2, 1, 64, 192, 4, seed=63)
    srcs = [enc.eval()(img.cuda()) for enc, _, _ in pairs]
    osrcs = [O.encoder_forward(img, sd, cfg, False) for _, _, sd in pairs]
    models = [dec.eval() for _, dec, _ in pairs]
    steps = 6
    manager = _manager(table)
    manager.reset(sequence_length=steps)
    target = torch.LongTensor(2).fill_(models[0].decoder.st_id).to("cuda")
    out = []
    for _ in range(steps):
        acc = None
        for m, model in enumerate(models):
            pr = F.softmax(model.step_forward(srcs[m], target).squeeze(), dim=-1)
            acc = pr if acc is None else acc + pr
        acc = acc / len(models)
        target, acc = manager.sift(acc)
        out.append(acc)
    got = torch.stack(out, 1)
    # oracle: same loop
    feats = [[None] * cfg["dec_layers"] for _ in pairs]
    state = O.sift_new_state(2, table)
    tgt = torch.full((2,), O.SOS_ID, dtype=torch.int64)
    ref = []
    for t in range(steps):
        acc = None
        for m, (_, _, sd) in enumerate(pairs):
            pr = F.softmax(O.decoder_step(tgt, t, feats[m], osrcs[m], sd, cfg)[:, 0], dim=-1)
            acc = pr if acc is None else acc + pr
        tgt, pr = O.sift(acc / len(pairs), state, table)
        ref.append(pr)
    ref = torch.stack(ref, 1)
    assert torch.equal(got.argmax(-1).cpu(), ref.argmax(-1))
    assert relerr(got, ref) < 1e-4
