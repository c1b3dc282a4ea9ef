"""Knowledge-distillation loss: oracle vs the reference's own loss_fn_kd vectors (CPU), the fused HIP
kernel vs both (GPU), and one student/teacher distillation step through the module interface (GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import satrn_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kd.npz")


def _inputs(B, T, V, seed):  # must match tests/golden/make_golden_kd.py
    s = O.det_tensor((B, T, V), seed, 4.0)
    t = O.det_tensor((B, T, V), seed + 1, 6.0)
    lab = (O.det_tensor((B, T), seed + 2, 1.0).abs() * 1e4).long() % V
    lab[:, -2:] = O.PAD_ID
    return s, t, lab


@pytest.mark.parametrize("case", ["a", "b"])
def test_oracle_kd_loss_matches_reference_vectors(case):
    z = np.load(GOLD)
    B, T, V, seed, temp = (int(v) for v in z[case + "_meta"])
    alpha = float(z[case + "_alpha"])
    s, t, lab = _inputs(B, T, V, seed)
    s.requires_grad_(True)
    loss = O.loss_fn_kd(s.transpose(1, 2), lab, t.transpose(1, 2), T=temp, alpha=alpha)
    loss.backward()
    assert abs(loss.item() - float(z[case + "_loss"])) < 1e-6
    assert np.allclose(s.grad.numpy(), z[case + "_grad"], rtol=0, atol=1e-8)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["a", "b"])
def test_device_kd_loss_matches_reference_vectors(case):
    import satrn_amd
    z = np.load(GOLD)
    B, T, V, seed, temp = (int(v) for v in z[case + "_meta"])
    alpha = float(z[case + "_alpha"])
    s, t, lab = _inputs(B, T, V, seed)
    sd = s.cuda().requires_grad_(True)
    loss = satrn_amd.loss_fn_kd(sd.transpose(1, 2), lab.cuda(), t.cuda().transpose(1, 2), T=temp, alpha=alpha)
    loss.backward()
    assert abs(loss.item() - float(z[case + "_loss"])) < 2e-5 * max(1.0, abs(float(z[case + "_loss"])))
    g, ref = sd.grad.cpu().numpy(), z[case + "_grad"]
    assert np.abs(g - ref).max() < 1e-6 + 1e-5 * np.abs(ref).max()


@pytest.mark.gpu
def test_distillation_step_through_the_module_interface():
    """train_distillation.py:88-131: student TF forward, teacher greedy forward (no grad), loss_fn_kd, backward, clip, step."""
    import satrn_amd
    from tests.test_model_gpu import build
    cfg = dict(O.CFG_LITE)
    student, ssd = build(cfg, 64, 192, "f32", 31)
    teacher, tsd = build(cfg, 64, 192, "f32", 32)
    img, expected = O.det_inputs(2, 1, 64, 192, 6, seed=80, pad_tail=1)
    imgd, expd = img.cuda(), expected.cuda()
    student.train(); teacher.eval()
    out = student(imgd, expd, True, 1.0).transpose(1, 2)
    with torch.no_grad():
        tout = teacher(imgd, expd, False, 0.0).transpose(1, 2)
    loss = satrn_amd.loss_fn_kd(outputs=out, labels=expd[:, 1:], teacher_outputs=tout)
    params = list(student.encoder.parameters()) + list(student.decoder.parameters())
    student.zero_grad()
    loss.backward()
    gn = torch.nn.utils.clip_grad_norm_(params, 2.0)
    # oracle: same graph in fp32 PyTorch
    osd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in ssd.items()}
    ologits = O.model_forward(img, expected, osd, cfg, True, True, O._BNState())
    with torch.no_grad():
        tlog = O.model_forward(img, expected, tsd, cfg, False)
    oloss = O.loss_fn_kd(ologits.transpose(1, 2), expected[:, 1:], tlog.transpose(1, 2))
    names = O.trainable_names(cfg)
    ograds = torch.autograd.grad(oloss, [osd[n] for n in names], allow_unused=True)
    assert abs(loss.item() - oloss.item()) < 1e-4 * max(1.0, abs(oloss.item()))
    og = dict(zip(names, ograds))
    w = "decoder.generator.weight"
    got = dict(student.named_parameters())[w].grad.cpu()
    ref = og[w] * min(1.0, 2.0 / (torch.sqrt(sum((g ** 2).sum() for g in ograds if g is not None)).item() + 1e-6))
    assert (got - ref).abs().max().item() < 1e-3 * ref.abs().max().item() + 1e-7
    assert gn.item() > 0
