"""GPU: the pipelined weight-stationary greedy decoder (bf16; one persistent workgroup per decoder ROLE with its weight slice
resident in LDS, images flowing through tagged-granule mailboxes).

Parity is pinned over the WHOLE decode by forced replay (satrn_model_greedy_forced): the CPU oracle's own greedy ids
(oracle.decoder_greedy_forward, the restatement of networks/EfficientSATRN.py:528-561) are fed to the HIP decoders, so every one of
the 231 steps sees the oracle's inputs and its logits are compared with the oracle's at a stated bf16 bound -- a mailbox-reuse or
history-index bug at step 200 fails here.  A free-running decode can only be followed up to its first near-tie (a token decided
inside bf16 noise legitimately changes everything after it); those tests stay, on top.

Every test asserts WHICH kernel produced the result (satrn_model_last_decode_path): a pipeline that gave up and was re-run on
the per-image kernel fails the test instead of passing on the fallback."""
import os
from satrn_amd import switches as sw

import numpy as np
import pytest
import torch

from oracle import satrn_oracle as O
from tests.test_model_gpu import build

pytestmark = pytest.mark.gpu

# bf16 bounds of a forced replay against the f32 CPU oracle (CFG_EFF, 128x384, random-init weights, eval mode).  The error is
# bf16 storage of activations / weights through 40 backbone blocks + 2 encoder + 3 decoder layers; it does not grow with the step
# index.  Measured on MI355X (gpurun_out/r3a/pytest.log): over all 8 x 231 steps max 8.8e-3 of max |logit| (first quarter 8.8e-3,
# last quarter 7.7e-3), rel-L2 5.0e-3; the two HIP kernels differ by 5.3e-3 at most; the f32 per-image kernel by 3e-6 absolute.
BF16_MAX_REL = 0.02      # max |logit - oracle| over every (image, step, class) / max |oracle logit|
BF16_L2_REL = 0.012      # ||logits - oracle|| / ||oracle|| over the whole decode
# between the two HIP kernels (same bf16 inputs, different summation grouping)
KERNEL_MAX_REL = 0.012


def _greedy(model, img, steps, pipe, forced=None):
    if pipe:
        sw.on("decode_pipe")
    else:
        sw.off("decode_pipe")
    try:
        lg, ids = model.greedy(img, steps, forced=forced)
        torch.cuda.synchronize()
        path, giveups, note = model.last_decode_path()
        assert giveups == 0, f"a pipelined decode gave up: {note}"
        assert path == ("pipe" if pipe else "per_image"), f"decoder path {path!r} ({note})"
        return lg.clone(), ids.clone()
    finally:
        sw.on("decode_pipe")


def _oracle_decode(cfg, sd, img, steps):
    with torch.no_grad():
        src = O.encoder_forward(img, sd, cfg, False)
        return O.decoder_greedy_forward(src, steps, sd, cfg)


@pytest.mark.parametrize("B,steps", [(8, 231), (3, 40)])
def test_forced_replay_of_oracle_ids_every_step(B, steps):
    """oracle ids -> pipeline AND per-image kernel (bf16) and the per-image kernel in f32: every step's logits against the oracle's"""
    cfg = dict(O.CFG_EFF)
    img, _ = O.det_inputs(B, 1, 128, 384, 4, seed=300 + B)
    model, sd = build(cfg, 128, 384, "bf16", 6)
    model.eval()
    ologits, oids = _oracle_decode(cfg, sd, img, steps)
    scale = ologits.abs().max().item()
    top2 = torch.topk(ologits, 2, dim=-1).values
    margin = top2[..., 0] - top2[..., 1]
    imgd = img.cuda()
    errs = {}
    for name, pipe in (("pipe", True), ("per_image", False)):
        lg, ids = _greedy(model, imgd, steps, pipe, forced=oids)
        lg = lg.cpu()
        assert torch.isfinite(lg).all()
        d = (lg - ologits).abs()
        per_step = d.amax(dim=(0, 2)) / scale           # [steps]
        l2 = ((lg - ologits).norm() / ologits.norm()).item()
        errs[name] = per_step.max().item()
        print(f"[forced {name} B={B} T={steps}] max rel err {per_step.max().item():.3e} (first quarter {per_step[: steps // 4 + 1].max().item():.3e}, "
              f"last quarter {per_step[-(steps // 4 + 1):].max().item():.3e}), rel-L2 {l2:.3e}")
        assert per_step.max().item() < BF16_MAX_REL, f"{name}: step {int(per_step.argmax())} is {per_step.max().item():.3e} of max |logit| away from the oracle"
        assert l2 < BF16_L2_REL
        # argmax of EVERY step (not a prefix): equal to the oracle's wherever the oracle's top-1 / top-2 margin is clear of the bound
        clear = margin > 2.5 * BF16_MAX_REL * scale
        assert (ids.cpu()[clear] == oids[clear]).all(), f"{name}: argmax differs from the oracle at a clear margin"
        print(f"[forced {name}] ids equal to the oracle's at {int((ids.cpu() == oids).sum())} of {B * steps} steps ({int(clear.sum())} with a clear margin)")
        if name == "pipe":
            lg_pipe = lg
        else:
            dk = (lg - lg_pipe).abs().max().item() / scale
            print(f"[forced] pipeline vs per-image kernel, every step: max rel diff {dk:.3e}")
            assert dk < KERNEL_MAX_REL
    # the forced mode itself, in the parity dtype: f32 per-image kernel, 1e-3 of the north star on every step, ids exact
    m32, sd32 = build(cfg, 128, 384, "f32", 6)
    m32.eval()
    lg32, ids32 = m32.greedy(imgd, steps, forced=oids)
    assert m32.last_decode_path()[0] == "per_image"
    e32 = (lg32.cpu() - ologits).abs().max().item()
    print(f"[forced f32 B={B} T={steps}] max abs err {e32:.3e} (max |logit| {scale:.2f})")
    assert e32 < 1e-3 * max(1.0, scale)
    assert (ids32.cpu()[margin > 1e-3] == oids[margin > 1e-3]).all()


@pytest.mark.parametrize("B,steps", [(2, 7), (5, 33), (64, 231), (112, 60)])
def test_pipelined_decoder_matches_per_image_decoder(B, steps):
    cfg = dict(O.CFG_EFF)
    model, sd = build(cfg, 128, 384, "bf16", 6)
    model.eval()
    img, _ = O.det_inputs(B, 1, 128, 384, 4, seed=90 + B)
    imgd = img.cuda()
    lg0, ids0 = _greedy(model, imgd, steps, pipe=False)
    scale = max(1.0, lg0.abs().max().item())
    # (1) the pipeline replaying the per-image kernel's ids: EVERY step of every image within the kernel-to-kernel bound
    lgf, idsf = _greedy(model, imgd, steps, pipe=True, forced=ids0)
    assert torch.isfinite(lgf).all()
    dmax = (lgf - lg0).abs().amax(dim=(0, 2)) / scale
    print(f"[pipe B={B} T={steps}] forced replay of the per-image ids: max rel diff {dmax.max().item():.3e} at step {int(dmax.argmax())}")
    assert dmax.max().item() < KERNEL_MAX_REL
    top2 = torch.topk(lg0, 2, dim=-1).values
    margin = (top2[..., 0] - top2[..., 1])
    clear = margin > 2.5 * KERNEL_MAX_REL * scale
    assert (idsf[clear] == ids0[clear]).all()
    # (2) free running: identical until a token is decided inside bf16 noise
    lg1, ids1 = _greedy(model, imgd, steps, pipe=True)
    assert torch.isfinite(lg1).all()
    agree = 0
    for b in range(B):
        same = (ids0[b] == ids1[b]).int()
        first_diff = int(same.argmin().item()) if same.min().item() == 0 else steps
        agree += first_diff
        if first_diff < steps:
            assert margin[b, first_diff].item() < 2.5 * KERNEL_MAX_REL * scale, f"image {b} step {first_diff}: ids differ at a clear margin {margin[b, first_diff].item():.3f}"
        if first_diff > 0:
            assert (lg1[b, :first_diff] - lg0[b, :first_diff]).abs().max().item() < KERNEL_MAX_REL * scale
    print(f"[pipe B={B} T={steps}] free-running identical prefix: {agree} of {B * steps} tokens")
    # run-to-run determinism of the pipeline (fixed-order partial sums): bit-equal logits
    lg2, ids2 = _greedy(model, imgd, steps, pipe=True)
    assert torch.equal(lg1, lg2) and torch.equal(ids1, ids2)


def test_pipelined_decoder_second_call_and_growing_batch():
    """mailboxes are re-zeroed per call; a second decode of another batch size reuses the model"""
    cfg = dict(O.CFG_EFF)
    model, sd = build(cfg, 128, 384, "bf16", 7)
    model.eval()
    for B, steps in ((3, 9), (8, 12), (3, 9)):
        img, _ = O.det_inputs(B, 1, 128, 384, 4, seed=70 + B)
        lg0, ids0 = _greedy(model, img.cuda(), steps, pipe=False)
        lg1, ids1 = _greedy(model, img.cuda(), steps, pipe=True, forced=ids0)
        assert torch.isfinite(lg1).all()
        assert (lg1 - lg0).abs().max().item() < KERNEL_MAX_REL * max(1.0, lg0.abs().max().item())


def test_decode_path_is_reported():
    """shapes outside the pipeline say so (path + note) instead of silently running another kernel"""
    cfg = dict(O.CFG_EFF)
    model, sd = build(cfg, 128, 384, "f32", 8)
    model.eval()
    img, _ = O.det_inputs(2, 1, 128, 384, 4, seed=5)
    model.greedy(img.cuda(), 5)
    path, giveups, note = model.last_decode_path()
    assert path == "per_image" and giveups == 0 and "shape" in note
    sw.off("decode_pipe")
    try:
        mb, _ = build(cfg, 128, 384, "bf16", 8)
        mb.eval()
        mb.greedy(img.cuda(), 5)
        path, giveups, note = mb.last_decode_path()
        assert path == "per_image" and "decode_pipe" in note
    finally:
        sw.on("decode_pipe")
    mb.greedy(img.cuda(), 5)
    assert mb.last_decode_path()[0] == "pipe"


def test_pipelined_decoder_with_decoding_manager_rules():
    """DecodingManager rules inside the pipeline's generator role (the reference's default at inference, inference.py:48):
    same masked probabilities / ids as the per-image kernel's sift wherever the masked top-1 / top-2 margin is clear, and the
    SAME entries are exactly zero (the blacklist depends on the ids only)."""
    from tests.test_rules_gpu import _manager, GOLD
    table = np.load(GOLD)["table"]
    cfg = dict(O.CFG_EFF)
    V = len(table) - 8
    cfg["num_classes"] = V
    model, sd = build(cfg, 128, 384, "bf16", 8)
    model.eval()
    model.decoder.manager = _manager(table)
    B, steps = 12, 40
    img, _ = O.det_inputs(B, 1, 128, 384, 4, seed=55)
    pr0, ids0 = _greedy(model, img.cuda(), steps, pipe=False)
    pr1, ids1 = _greedy(model, img.cuda(), steps, pipe=True)
    assert torch.isfinite(pr1).all()
    assert (pr1.sum(-1) <= 1.0 + 1e-3).all() and (pr1 >= 0).all()
    top2 = torch.topk(pr0, 2, dim=-1).values
    margin = top2[..., 0] - top2[..., 1]
    agree = 0
    for b in range(B):
        same = (ids0[b] == ids1[b]).int()
        first_diff = int(same.argmin().item()) if same.min().item() == 0 else steps
        agree += first_diff
        if first_diff < steps:
            assert margin[b, first_diff].item() < 2e-2, f"image {b} step {first_diff}: ids differ at a clear margin"
        n = min(first_diff + 1, steps)   # the blacklist of step t depends on the ids before t
        assert ((pr0[b, :n] == 0) == (pr1[b, :n] == 0)).all(), f"image {b}: different entries blacklisted"
        if first_diff > 0:
            assert (pr1[b, :first_diff] - pr0[b, :first_diff]).abs().max().item() < 2e-2
    print(f"[pipe + rules] identical prefix: {agree} of {B * steps} tokens")
    assert agree >= 0.5 * B * steps
