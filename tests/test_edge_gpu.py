"""GPU: edge cases of the module interface (shapes, degenerate batches, gradient-accumulation semantics, checkpoints)."""
import numpy as np
import pytest
import torch

from oracle import satrn_oracle as O
from tests.test_model_gpu import build, relerr

pytestmark = pytest.mark.gpu


def test_batch_of_one_and_single_step():
    """B=1 (the reference's greedy loop breaks on target.squeeze() for B=1; the engine has no such limit) and L=2 (T=1)."""
    cfg = dict(O.CFG_LITE)
    model, sd = build(cfg, 64, 192, "f32", 7)
    img, expected = O.det_inputs(1, 1, 64, 192, 1, seed=50)
    model.eval()  # before the training forward below moves the BN running statistics away from sd's
    lg, ids = model.greedy(img.cuda(), 5)
    src = O.encoder_forward(img, sd, cfg, False)
    olg, oids = O.decoder_greedy_forward(src, 5, sd, cfg)
    assert (ids.cpu() == oids).all() and relerr(lg, olg) < 1e-4
    model.train()
    logits = model(img.cuda(), expected.cuda(), True, 1.0)
    loss = model.criterion(logits.transpose(1, 2), expected.cuda()[:, 1:])
    loss.backward()
    oloss, ologits, _, _ = O.forward_backward(img, expected, sd, cfg)
    assert logits.shape == (1, 1, 245) and abs(loss.item() - oloss.item()) < 1e-4


def test_all_pad_row_and_fully_padded_batch():
    """a sample whose targets are all PAD contributes nothing; a batch with no valid token gives loss 0 (not NaN) here
    (the reference's mean over zero tokens is NaN -- documented difference, gradients are zero either way)."""
    cfg = dict(O.CFG_LITE)
    model, sd = build(cfg, 64, 192, "f32", 8)
    img, expected = O.det_inputs(3, 1, 64, 192, 6, seed=51)
    expected[1, 1:] = O.PAD_ID
    model.train()
    logits = model(img.cuda(), expected.cuda(), True, 1.0)
    loss = model.criterion(logits.transpose(1, 2), expected.cuda()[:, 1:])
    model.zero_grad()
    loss.backward()
    oloss, _, ograds, _ = O.forward_backward(img, expected, sd, cfg)
    assert abs(loss.item() - oloss.item()) < 1e-4
    g = dict(model.named_parameters())["decoder.generator.weight"].grad.cpu()
    assert relerr(g, ograds["decoder.generator.weight"]) < 1e-3
    expected[:, 1:] = O.PAD_ID
    logits = model(img.cuda(), expected.cuda(), True, 1.0)
    loss = model.criterion(logits.transpose(1, 2), expected.cuda()[:, 1:])
    assert loss.item() == 0.0


def test_rgb_input_and_other_resolution():
    """FLAGS.data.rgb = 3 (the reference's YAML default) and a 64x256 input whose feature map goes 31x127 -> ... -> 2x8
    through the TF-SAME strided convolutions."""
    cfg = dict(O.CFG_EFF, rgb=3)
    model, sd = build(cfg, 64, 256, "f32", 9)
    img, expected = O.det_inputs(2, 3, 64, 256, 5, seed=52, pad_tail=1)
    model.train()
    logits = model(img.cuda(), expected.cuda(), True, 1.0)
    oloss, ologits, _, _ = O.forward_backward(img, expected, sd, cfg)
    assert relerr(logits, ologits) < 1e-3


def test_gradient_accumulation_and_zero_grad_semantics():
    """two backward() calls without zero_grad accumulate (like autograd); zero_grad(set_to_none) starts from zero."""
    cfg = dict(O.CFG_LITE)
    model, sd = build(cfg, 64, 192, "f32", 10)
    img, expected = O.det_inputs(2, 1, 64, 192, 6, seed=53)
    imgd, expd = img.cuda(), expected.cuda()
    model.train()

    def fb():
        logits = model(imgd, expd, True, 1.0)
        model.criterion(logits.transpose(1, 2), expd[:, 1:]).backward()

    model.zero_grad()
    fb()
    w = dict(model.named_parameters())["decoder.generator.weight"]
    g1 = w.grad.detach().clone()
    # BN running stats changed, but gradients do not depend on them in train mode
    fb()
    g2 = w.grad.detach().clone()
    assert relerr(g2, 2 * g1) < 1e-4
    model.zero_grad()
    fb()
    assert relerr(w.grad, g1) < 1e-4


def test_torch_optimizer_on_flat_parameter_views_matches_fused_step():
    """nn.Parameters are views of one flat buffer: clip_grad_norm_ + torch.optim.AdamW on them == the fused clip+AdamW
    kernel fed the same gradients (phase 2 of train_step).  Gradients are shared rather than recomputed because Adam's
    first steps are lr*sign(g): the atomics' rounding noise on exactly-zero true gradients would flip whole steps."""
    cfg = dict(O.CFG_LITE)
    img, expected = O.det_inputs(2, 1, 64, 192, 6, seed=54)
    imgd, expd = img.cuda(), expected.cuda()
    a, _ = build(cfg, 64, 192, "f32", 11)
    b, _ = build(cfg, 64, 192, "f32", 11)
    a.train(); b.train()
    params = list(a.encoder.parameters()) + list(a.decoder.parameters())
    opt = torch.optim.AdamW(params, lr=5e-4, weight_decay=1e-6)
    for _ in range(3):
        logits = a(imgd, expd, True, 1.0)
        loss = a.criterion(logits.transpose(1, 2), expd[:, 1:])
        opt.zero_grad()
        loss.backward()
        b.flat_grad().copy_(a.flat_grad())
        torch.cuda.synchronize()
        torch.nn.utils.clip_grad_norm_(params, 2.0)
        opt.step()
        b.train_step(imgd, expd, 5e-4, phase=2)
        torch.cuda.synchronize()
    pa, pb = a.flat_params().detach().cpu(), b.flat_params().detach().cpu()
    upd = (pa - pb).abs().max().item()
    print("max param diff after 3 steps:", upd)
    assert upd < 2e-6  # steps are ~5e-4


def test_dual_optimizer_step_matches_two_torch_optimizers():
    """train_modules/train_dual_opt.py:87-113: clip_grad_norm_ per group (encoder / decoder) and two Adam optimizers with
    their own learning rates == the fused dual step fed the same gradients."""
    cfg = dict(O.CFG_LITE)
    img, expected = O.det_inputs(2, 1, 64, 192, 6, seed=57)
    imgd, expd = img.cuda(), expected.cuda()
    a, _ = build(cfg, 64, 192, "f32", 13)
    b, _ = build(cfg, 64, 192, "f32", 13)
    a.train(); b.train()
    enc_p, dec_p = list(a.encoder.parameters()), list(a.decoder.parameters())
    enc_opt, dec_opt = torch.optim.Adam(enc_p, lr=1e-4), torch.optim.Adam(dec_p, lr=5e-4)
    for it in range(3):
        logits = a(imgd, expd, True, 1.0)
        loss = a.criterion(logits.transpose(1, 2), expd[:, 1:])
        enc_opt.zero_grad(); dec_opt.zero_grad()
        loss.backward()
        b.flat_grad().copy_(a.flat_grad())
        torch.cuda.synchronize()
        # a clip value that bites for one group only on some iterations
        mg = 0.05 if it == 1 else 2.0
        en = torch.nn.utils.clip_grad_norm_(enc_p, mg)
        dn = torch.nn.utils.clip_grad_norm_(dec_p, mg)
        enc_opt.step(); dec_opt.step()
        b.train_step(imgd, expd, (1e-4, 5e-4), weight_decay=0.0, max_grad_norm=mg, phase=2)
        ge, gd = b.read_grad_norms()
        assert abs(ge - en.item()) < 1e-4 * max(1.0, en.item()) and abs(gd - dn.item()) < 1e-4 * max(1.0, dn.item())
    pa, pb = a.flat_params().detach().cpu(), b.flat_params().detach().cpu()
    upd = (pa - pb).abs().max().item()
    print("max param diff after 3 dual steps:", upd)
    assert upd < 2e-6
    with pytest.raises(ValueError):
        b.train_step(imgd, expd, (1e-4, 5e-4), use_graph=True)


def test_last_sequence_is_the_argmax_of_the_training_logits():
    """train_modules/train_single_opt.py:80-84: `sequence` for the per-step metrics, after a fused step, without logits."""
    import satrn_amd
    cfg = dict(O.CFG_LITE)
    img, expected = O.det_inputs(3, 1, 64, 192, 9, seed=58, pad_tail=2)
    imgd, expd = img.cuda(), expected.cuda()
    a, _ = build(cfg, 64, 192, "f32", 14)
    a.train()
    logits = a(imgd, expd, True, 1.0).detach()
    B, L = expd.shape
    seq = a.last_sequence(B, L)                       # after a module-API forward
    assert seq.shape == (B, L - 1) and (seq == logits.argmax(-1)).all()
    a.train_step(imgd, expd, 5e-4, phase=1)           # same weights, same batch statistics, dropout 0 -> same logits
    seq2 = a.last_sequence(B, L)
    assert (seq2 == seq).all()
    a.train_step(imgd, expd, 5e-4)                    # a full fused step keeps the step's logits readable
    assert (a.last_sequence(B, L) == seq).all()
    a.greedy(imgd, 4)                                 # any other run of the model ends that
    with pytest.raises(satrn_amd.SatrnError):
        a.last_sequence(B, L)


def test_state_dict_roundtrip_on_device(tmp_path):
    cfg = dict(O.CFG_LITE)
    a, sd = build(cfg, 64, 192, "bf16", 12)
    img, expected = O.det_inputs(2, 1, 64, 192, 6, seed=55)
    a.train()
    a.train_step(img.cuda(), expected.cuda(), 1e-3)
    path = tmp_path / "ckpt.pth"
    torch.save({"model": a.state_dict()}, path)
    import satrn_amd
    from tests.test_model_gpu import make_flags, _DS
    b = satrn_amd.LiteSATRN(make_flags(cfg, 64, 192), _DS(), torch.load(path)["model"], dtype="bf16").to("cuda")
    a.eval(); b.eval()
    la, ia = a.greedy(img.cuda(), 6)
    lb, ib = b.greedy(img.cuda(), 6)
    assert torch.equal(ia, ib) and torch.equal(la, lb)
    assert int(dict(a.named_buffers())["encoder.shallow_cnn.batch_norm0.num_batches_tracked"]) == 1


def test_decode_length_limit_is_an_error():
    import satrn_amd
    cfg = dict(O.CFG_LITE)
    model, _ = build(cfg, 64, 192, "f32", 13)
    img, _ = O.det_inputs(1, 1, 64, 192, 2, seed=56)
    model.eval()
    with pytest.raises(satrn_amd.SatrnError):
        model.greedy(img.cuda(), 501)  # PositionEncoder1D(max_len=500), networks/EfficientSATRN.py:401


def test_eval_statistics_step_matches_oracle():
    """train_step(bn_eval=True): BatchNorm on its running statistics + no dropout, gradients recorded (module.eval()
    semantics) -- the per-sample-independent mode the data-parallel equivalence test relies on -- against the oracle."""
    cfg = dict(O.CFG_EFF)
    model, sd = build(cfg, 64, 96, "f32", 21, dropout=0.1)
    img, expected = O.det_inputs(3, 1, 64, 96, 7, seed=71, pad_tail=1)
    model.train()
    model.train_step(img.cuda(), expected.cuda(), 0.0, phase=1, bn_eval=True)
    torch.cuda.synchronize()
    oloss, _, ograds, _ = O.forward_backward(img, expected, sd, cfg, bn_train=False)
    loss = model.read_loss()[0]
    assert abs(loss - oloss.item()) < 1e-4
    model._attach_grads()   # train_step works on the flat buffer; give the parameters their .grad views
    params = dict(model.named_parameters())
    gmax = max(g.abs().max().item() for g in ograds.values())
    worst = max((params[n].grad.detach().cpu() - g).abs().max().item() for n, g in ograds.items())
    print("eval-statistics step: worst abs grad err", worst, "of max", gmax)
    assert worst < 2e-3 * gmax
    # running statistics untouched
    for n, b in model.named_buffers():
        if n.endswith("running_mean") or n.endswith("running_var"):
            assert torch.equal(b.detach().cpu(), sd[n]), n


def test_workspace_regrow_keeps_optimizer_state():
    """The reference loader pads every batch to its own max length (data/loader.py:11): a batch longer than all earlier ones
    makes the engine take a bigger workspace.  Adam's moments, its step count and the dropout RNG must survive that
    (round-1 defect: they lived in the workspace and silently restarted)."""
    cfg = dict(O.CFG_LITE)
    short = O.det_inputs(2, 1, 64, 192, 6, seed=81)
    longer = O.det_inputs(3, 1, 64, 192, 14, seed=82)
    a, _ = build(cfg, 64, 192, "f32", 23)
    b, _ = build(cfg, 64, 192, "f32", 23)
    a.train(); b.train()
    a.reserve(3, 15, "cuda")          # pre-sized: never regrows
    ws_a = a._ws.data_ptr()
    for m in (a, b):
        for img, exp in (short, short, longer, longer, short):
            m.train_step(img.cuda(), exp.cuda(), 5e-4)
        torch.cuda.synchronize()
    assert a._ws.data_ptr() == ws_a and b._ws_key == (3, 15)
    sa, sb = a.optimizer_state_dict(), b.optimizer_state_dict()
    assert sa["step"] == 5 and sb["step"] == 5
    assert sa["rng"] == sb["rng"]
    assert relerr(sb["exp_avg"], sa["exp_avg"]) < 1e-3 and relerr(sb["exp_avg_sq"], sa["exp_avg_sq"]) < 1e-3
    assert sb["exp_avg"].abs().max().item() > 0
    frac = ((a.flat_params() - b.flat_params()).abs() > 1e-4).float().mean().item()
    assert frac < 0.01, f"{frac:.4f} of the parameters differ between the pre-sized and the regrown run"
    # round trip through the checkpoint interface
    c, _ = build(cfg, 64, 192, "f32", 23)
    c.reserve(3, 15, "cuda")
    c.load_state_dict(b.state_dict())
    c.load_optimizer_state_dict(sb)
    c.train()
    for m in (b, c):
        m.train_step(longer[0].cuda(), longer[1].cuda(), 5e-4)
    torch.cuda.synchronize()
    assert c.optimizer_state_dict()["step"] == 6
    frac = ((c.flat_params() - b.flat_params()).abs() > 1e-4).float().mean().item()
    assert frac < 0.01


def test_out_of_range_token_ids_are_flagged_not_followed():
    """collate pads `expected` with -1 (data/loader.py:12-16); if that reaches the model before the -1 -> PAD rewrite the
    embedding kernel must not read / atomicAdd out of bounds: the element is skipped, the device error word is set and
    read_loss / check_device_error raise (nn.Embedding raises an index error in the reference)."""
    import satrn_amd
    cfg = dict(O.CFG_LITE)
    model, _ = build(cfg, 64, 192, "f32", 24)
    img, expected = O.det_inputs(2, 1, 64, 192, 6, seed=83)
    bad = expected.clone()
    bad[1, 4:] = -1
    model.train()
    model.train_step(img.cuda(), bad.cuda(), 0.0, phase=1)
    with pytest.raises(satrn_amd.SatrnError, match="out of range"):
        model.read_loss()
    g = model.flat_grad()
    assert torch.isfinite(g).all()
    # the flag is cleared by the read; a clean batch then passes
    model.train_step(img.cuda(), expected.cuda(), 0.0, phase=1)
    assert model.read_loss()[0] > 0
    model.check_device_error()
    bad2 = expected.clone()
    bad2[0, 2] = 400
    logits = model(img.cuda(), bad2.cuda(), True, 1.0)
    with pytest.raises(satrn_amd.SatrnError, match="out of range"):
        model.check_device_error()
    assert torch.isfinite(logits).all()


def test_input_geometry_is_validated():
    import satrn_amd
    cfg = dict(O.CFG_LITE)
    model, _ = build(cfg, 64, 192, "f32", 25)
    img, expected = O.det_inputs(2, 1, 32, 192, 6, seed=84)
    with pytest.raises(satrn_amd.SatrnError, match="input must be"):
        model(img.cuda(), expected.cuda(), True, 1.0)
    with pytest.raises(satrn_amd.SatrnError, match="input must be"):
        model.greedy(img.cuda(), 4)


def test_parameter_edit_between_fused_steps_is_repacked():
    """load_state_dict / a manual edit between two train_step calls must reach the packed compute copies (round-1 defect:
    only the first step checked)."""
    cfg = dict(O.CFG_LITE)
    img, expected = O.det_inputs(2, 1, 64, 192, 6, seed=85)
    a, sd = build(cfg, 64, 192, "f32", 26)
    a.train()
    a.train_step(img.cuda(), expected.cuda(), 5e-4)
    sd2 = O.det_state_dict(cfg, 27)
    a.load_state_dict(sd2)
    a.train_step(img.cuda(), expected.cuda(), 0.0, phase=1)
    oloss, _, _, _ = O.forward_backward(img, expected, sd2, cfg)
    assert abs(a.read_loss()[0] - oloss.item()) < 1e-4
