"""GPU: size-independent properties at BASELINE.json's FULL sizes (configs[1]: EfficientSATRN, 32 x 1x128x384, T=128 for
training; config 5: 64 images x 231 greedy steps for decoding), where the CPU oracle would take minutes.  The oracle /
golden parity tests (tests/test_model_gpu.py) run the same code paths at sizes the oracle finishes in seconds."""
import os
import pytest
import torch
import torch.nn.functional as F
from satrn_amd import switches as sw

from oracle import satrn_oracle as O
from tests.test_model_gpu import build, relerr, _DS

pytestmark = pytest.mark.gpu
H, W, T = 128, 384, 128


def synth(B, seed):
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(B, 1, H, W, generator=g)
    exp = torch.randint(3, 245, (B, T + 1), generator=g)
    exp[:, 0] = 0
    exp[:, -1] = 1
    return img.cuda(), exp.cuda()


@pytest.fixture(scope="module")
def model32():
    model, _ = build(dict(O.CFG_EFF), H, W, "f32", 31)
    return model


def test_full_size_backward_is_linear_in_the_upstream_gradient_and_ce_matches_torch(model32):
    model = model32
    model.train()
    img, exp = synth(32, 1)
    logits = model(img, exp, True, 1.0)
    assert logits.shape == (32, T, 245)
    # fused CE path == torch's CE on the same logits (ignore_index = PAD), full size
    expp = exp.clone()
    expp[:, -20:] = O.PAD_ID
    lg = logits.detach().clone().requires_grad_(True)
    ref = F.cross_entropy(lg.transpose(1, 2), expp[:, 1:], ignore_index=O.PAD_ID)
    ref.backward()
    model.zero_grad()
    loss = model.criterion(logits.transpose(1, 2), expp[:, 1:])
    assert abs(loss.item() - ref.item()) < 1e-5 * max(1.0, abs(ref.item()))
    loss.backward()
    g1 = model.flat_grad().clone()
    assert torch.isfinite(g1).all() and g1.abs().max() > 0
    # linearity: the same recorded graph, upstream gradient scaled by 4 (a power of two: exact in every dtype)
    model.zero_grad()
    logits2 = model(img, exp, True, 1.0)
    assert torch.equal(logits2, logits)   # train-mode forward is a function of the batch only (dropout 0): same bits
    (4.0 * model.criterion(logits2.transpose(1, 2), expp[:, 1:])).backward()
    g4 = model.flat_grad()
    # f32 reductions are deterministic (fixed-order partial folds) and a power-of-two scale commutes with every rounding:
    # the scaled backward is the same bits times four
    assert torch.equal(g4, 4.0 * g1), f"max rel diff {relerr(g4, 4.0 * g1):.3e}"


def test_full_size_teacher_forced_logits_are_causal_in_the_tokens(model32):
    """order mask + pad mask (networks/EfficientSATRN.py:469-478): position t sees tokens <= t only; a PAD key is masked
    for every LATER query but the positions before it are untouched."""
    model = model32
    model.train()   # batch statistics: a function of the images only, so both runs normalise identically
    img, exp = synth(32, 2)
    a = model(img, exp, True, 1.0).detach().clone()
    exp2 = exp.clone()
    exp2[:, 70:] = torch.randint(3, 245, exp2[:, 70:].shape, device=exp.device)
    exp2[:, 100:] = O.PAD_ID
    b = model(img, exp2, True, 1.0).detach()
    # decoder input is expected[:, :-1]: logits at positions < 70 read tokens 0..69 only
    assert relerr(b[:, :70], a[:, :70]) < 1e-5
    assert relerr(b[:, 70:], a[:, 70:]) > 1e-3


def test_full_size_decode_rows_are_independent_and_beam1_is_greedy():
    """config 5 (64 x 231): every image decodes alone -- a sub-batch reproduces its rows of the full batch; best-first beam
    search with beam_width 1 walks the greedy chain (same decoder step, argmax of log-softmax = argmax of logits)."""
    model, _ = build(dict(O.CFG_EFF), H, W, "bf16", 32)
    model.eval()
    img, _ = synth(64, 3)
    logits, ids = model.greedy(img, 231)
    assert logits.shape == (64, 231, 245) and ids.shape == (64, 231)
    assert (logits.argmax(-1) == ids).all()
    pick = [5, 17, 63]
    l3, i3 = model.greedy(img[pick], 231)
    top2 = logits[pick].topk(2, -1).values
    sure = (top2[..., 0] - top2[..., 1]) > 5e-2     # bf16: the encoder's GEMM tiling depends on the batch
    assert relerr(l3, logits[pick]) < 3e-2
    # ids are fed back: compare up to the first unsure step of each row
    for r in range(len(pick)):
        ok = sure[r].cpu()
        first_unsure = int((~ok).nonzero()[0]) if (~ok).any() else 231
        assert (i3[r, :first_unsure] == ids[pick[r], :first_unsure]).all()

    class L:
        dataset = _DS()
    seq = model.beam_search(img, L, beam_width=1, max_sequence=231).cuda()
    assert (seq[:, 0] == O.SOS_ID).all()
    # the search kernel runs the per-image decoder step: it must walk the per-image greedy kernel's chain EXACTLY, and the
    # pipelined greedy decoder's chain (same mathematics, partial sums added in another order) up to its first near-tie
    sw.off("decode_pipe")
    try:
        _, ids_pi = model.greedy(img, 231)
    finally:
        sw.on("decode_pipe")
    margin = logits.topk(2, -1).values
    sure_all = ((margin[..., 0] - margin[..., 1]) > 5e-2).cpu()
    for b in range(64):
        row = ids_pi[b]
        eos = (row == O.EOS_ID).nonzero()
        n = int(eos[0]) + 1 if len(eos) else 230      # the search stops at the first <EOS>; 230 expansions at most
        n = min(n, 230)
        assert (seq[b, 1:1 + n] == row[:n]).all(), b
        assert (seq[b, 1 + n:] == O.PAD_ID).all(), b
        first_unsure = int((~sure_all[b]).nonzero()[0]) if (~sure_all[b]).any() else 231
        m = min(n, first_unsure)
        assert (seq[b, 1:1 + m] == ids[b, :m]).all(), b
