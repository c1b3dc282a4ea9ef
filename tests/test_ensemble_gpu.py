"""GPU: the step-wise ensemble interface -- EfficientSATRN_encoder.forward and
EfficientSATRN_decoder.step_forward / reset_status driven exactly like utils/ensemble_utils.py:70-103 drives them,
checked against the oracle's restatement of that loop."""
import pytest
import torch
import torch.nn.functional as F

from oracle import satrn_oracle as O
from tests.test_model_gpu import _DS, make_flags, relerr

pytestmark = pytest.mark.gpu

H, W = 64, 192


def _halves(cfg, wseed, dtype):
    import satrn_amd
    sd = O.det_state_dict(cfg, wseed)
    flags = make_flags(cfg, H, W)
    enc_sd = {k: v for k, v in sd.items() if k.startswith("encoder.")}
    dec_sd = {k: v for k, v in sd.items() if k.startswith("decoder.")}
    enc = satrn_amd.get_network("EfficientSATRN_encoder", flags, enc_sd, "cuda", _DS(), dtype=dtype)
    dec = satrn_amd.get_network("EfficientSATRN_decoder", flags, dec_sd, "cuda", _DS(), dtype=dtype)
    return enc, dec, sd


def test_halves_have_the_reference_state_dict_keys():
    cfg = dict(O.CFG_EFF)
    enc, dec, sd = _halves(cfg, 3, "f32")
    assert set(enc.state_dict().keys()) == {k for k in sd if k.startswith("encoder.")}
    assert set(dec.state_dict().keys()) == {k for k in sd if k.startswith("decoder.")}
    assert dec.decoder.st_id == 0 and dec.decoder.layer_num == cfg["dec_layers"]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_single_model_step_loop_matches_greedy_oracle(dtype):
    cfg = dict(O.CFG_EFF)
    enc, dec, sd = _halves(cfg, 4, dtype)
    img, _ = O.det_inputs(3, 1, H, W, 4, seed=60)
    enc.eval(); dec.eval()
    src = enc(img.cuda())
    osrc = O.encoder_forward(img, sd, cfg, False)
    tol = 1e-4 if dtype == "f32" else 6e-2
    assert src.shape == osrc.shape and relerr(src, osrc) < tol
    steps = 7
    # feed the oracle's own tokens so that a bf16 near-tie cannot fork the sequences: per-step logits stay comparable
    olog, oids = O.decoder_greedy_forward(osrc, steps, sd, cfg)
    for rep in range(2):  # the second pass checks reset_status()
        target = torch.full((3,), dec.decoder.st_id, dtype=torch.int64, device="cuda")
        outs = []
        for t in range(steps):
            out = dec.step_forward(osrc.cuda() if dtype == "f32" else src, target)
            assert out.shape == (3, 1, cfg["num_classes"])
            outs.append(out[:, 0])
            target = oids[:, t].cuda()
        got = torch.stack(outs, 1)
        assert relerr(got, olog) < (1e-4 if dtype == "f32" else 8e-2)
        if dtype == "f32":
            assert torch.equal(got.argmax(-1).cpu(), oids)
        assert dec.step_idx == steps
        dec.reset_status()
        assert dec.step_idx == 0


def test_two_model_ensemble_like_the_reference_driver():
    """utils/ensemble_utils.py:70-103: softmax-average over models, argmax, feed back; reset_status between batches."""
    cfg = dict(O.CFG_EFF)
    pairs = [_halves(cfg, 5, "f32"), _halves(cfg, 6, "f32")]
    img, _ = O.det_inputs(2, 1, H, W, 4, seed=61)
    srcs = [enc.eval()(img.cuda()) for enc, _, _ in pairs]
    osrcs = [O.encoder_forward(img, sd, cfg, False) for _, _, sd in pairs]
    steps = 6
    oprob, oids = O.ensemble_greedy_forward(osrcs, steps, [sd for _, _, sd in pairs], cfg)
    models = [dec.eval() for _, dec, _ in pairs]
    st_id = models[0].decoder.st_id
    target = torch.LongTensor(2).fill_(st_id).to("cuda")
    out = []
    for _ in range(steps):
        one_step_out = None
        for m, model in enumerate(models):
            _out = model.step_forward(srcs[m], target)
            if _out.ndim > 2:
                _out = _out.squeeze()
            assert _out.ndim == 2
            one_step_out = F.softmax(_out, dim=-1) if one_step_out is None else one_step_out + F.softmax(_out, dim=-1)
        one_step_out = one_step_out / len(models)
        target = torch.argmax(one_step_out, dim=-1)
        out.append(one_step_out)
    got = torch.stack(out, dim=1)
    for model in models:
        model.reset_status()
    assert torch.equal(got.argmax(-1).cpu(), oids)
    assert relerr(got, oprob) < 1e-4


def test_stale_session_and_exhaustion_are_errors():
    import satrn_amd
    cfg = dict(O.CFG_EFF)
    enc, dec, sd = _halves(cfg, 7, "f32")
    dec.max_steps = 2
    img, _ = O.det_inputs(2, 1, H, W, 4, seed=62)
    src = enc.eval()(img.cuda())
    tgt = torch.zeros(2, dtype=torch.int64, device="cuda")
    dec.step_forward(src, tgt)
    dec.step_forward(src, tgt)
    with pytest.raises(satrn_amd.SatrnError):
        dec.step_forward(src, tgt)  # third step of a 2-step session
    dec.reset_status()
    dec.step_forward(src, tgt)
    dec._full.encode(img.cuda())  # any other call on the engine model ends the session
    with pytest.raises(satrn_amd.SatrnError):
        dec.step_forward(src, tgt)
