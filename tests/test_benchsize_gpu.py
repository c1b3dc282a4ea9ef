"""GPU: the BENCHMARKED combination itself -- EfficientSATRN, bf16, 32 x 1x128x384, T = 128 (BASELINE.json configs[1]) -- asserted,
not only timed.  The golden / oracle tests run at B = 2, where no product reaches the persistent kernels' size thresholds; here the
bf16 engine is held against the f32 engine (the mode pinned to the reference at 1e-3) on the same weights and batch, and the route
counters prove that the large-shape kernels (persistent GEMM / shifted-GEMM convolution / persistent weight gradient, the one-launch
BatchNorm + squeeze-and-excite) ran INSIDE the model."""
import ctypes
import re

import pytest
import torch

import bench

pytestmark = pytest.mark.gpu
H, W, T, B = 128, 384, 128, 32


def _routes(reset=False):
    import satrn_amd
    lib = satrn_amd._lib.load()
    out = (ctypes.c_longlong * 16)()
    n = lib.satrn_route_counts(out, 16, int(reset))
    assert n >= 10
    return dict(gemm_big=out[0], gemm_big_conv=out[1], wgrad_big=out[2], gemm_tile=out[3], wgrad_tile=out[4], bn_pool_se=out[5], mbconv_fwd=out[6],
                mbconv_bwd=out[7], gemm_tall=out[8], ar_fused=out[9])


def _device_error():
    import satrn_amd
    return satrn_amd._lib.load().satrn_device_error(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))


def _stage(name):
    m = re.match(r"encoder\.shallow_cnn\.eff_block\.(\d+)\.", name)
    if m:
        return "stage" + m.group(1)
    if name.startswith("encoder.shallow_cnn."):
        return "stem_last"
    if name.startswith("encoder."):
        return "encoder"
    m = re.match(r"decoder\.attention_layers\.(\d+)\.", name)
    return "dec_layer" + m.group(1) if m else "decoder_other"


def _run(dtype, img, exp, bn_eval):
    torch.manual_seed(21)
    m = bench.make_model(dtype, H, W, 0.0).to(img.device)
    m.train()
    _routes(reset=True)
    if bn_eval:
        # module.eval() semantics WITH gradients (running statistics, no dropout): the smooth reference point, batch statistics out of
        # the picture; phase 1 = forward + CE + backward, no optimizer
        m.train_step(img, exp, 0.0, phase=1, bn_eval=True)
        torch.cuda.synchronize()
        loss = m.read_loss()[0]      # raises on a non-zero device error word
        logits = None
    else:
        logits = m(img, exp, True, 1.0)
        loss_t = m.criterion(logits.transpose(1, 2), exp[:, 1:])
        m.zero_grad()
        loss_t.backward()
        torch.cuda.synchronize()
        loss = float(loss_t.item())
        logits = logits.detach().float().clone()
    routes = _routes()
    assert _device_error() == 0
    grads = {n: p.grad.detach().float().clone() for n, p in m.named_parameters()} if not bn_eval else None
    return dict(loss=loss, logits=logits, flat=m.flat_grad().detach().float().clone(), grads=grads, routes=routes)


def _rel(a, b):
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _cos(a, b):
    return (torch.dot(a.flatten(), b.flatten()) / (a.norm() * b.norm()).clamp_min(1e-30)).item()


def test_benchmark_configuration_bf16_against_f32_bn_eval():
    """(a) running-statistics mode: every sample independent of its batch, no ReLU / BatchNorm mask amplification -- bf16 and f32
    agree to 9e-2 rel-L2 / cosine 0.996 on the flat gradient (measured this round; training mode: 0.40 / 0.92)."""
    img, exp = bench.synth(B, H, W, T, 21, torch.device("cuda:0"))
    f = _run("f32", img, exp, True)
    b = _run("bf16", img, exp, True)
    err, cs = _rel(b["flat"], f["flat"]), _cos(b["flat"], f["flat"])
    print(f"[B=32 bn_eval] loss f32 {f['loss']:.5f} bf16 {b['loss']:.5f}; flat gradient rel-L2 {err:.3e} cosine {cs:.5f}; routes {b['routes']}")
    assert abs(b["loss"] - f["loss"]) < 1e-2 * max(1.0, abs(f["loss"]))
    assert err < 0.13 and cs > 0.992
    assert torch.isfinite(b["flat"]).all()


def test_benchmark_configuration_bf16_against_f32_train_mode_and_routes():
    """(b) training mode (batch statistics), dropout off: logits within 5 %, loss within 1e-3 relative, per-stage weight-gradient cosines
    not below what profiles/r03_bf16_grad_error.txt recorded minus a margin (backbone stages 0.866-0.875 -> 0.80, decoder layers
    0.99+ -> 0.97); (c) the device error word is clear and the large-shape routes were taken by BOTH passes of the bf16 model."""
    img, exp = bench.synth(B, H, W, T, 21, torch.device("cuda:0"))
    f = _run("f32", img, exp, False)
    b = _run("bf16", img, exp, False)
    lerr = ((b["logits"] - f["logits"]).abs().max() / f["logits"].abs().max()).item()
    gerr, gcos = _rel(b["flat"], f["flat"]), _cos(b["flat"], f["flat"])
    print(f"[B=32 train] loss f32 {f['loss']:.5f} bf16 {b['loss']:.5f}; logits rel err {lerr:.3e}; flat gradient rel-L2 {gerr:.3f} cosine {gcos:.4f}")
    assert lerr < 5e-2
    assert abs(b["loss"] - f["loss"]) < 1e-3 * abs(f["loss"])
    assert gcos > 0.88 and gerr < 0.5      # measured 0.918 / 0.404 (rounds 2-3); a regression of the bf16 path moves these first
    groups = {}
    for n, gf in f["grads"].items():
        g = groups.setdefault(_stage(n), [0.0, 0.0, 0.0])
        gb = b["grads"][n]
        g[0] += (gb * gf).sum().item(); g[1] += gf.pow(2).sum().item(); g[2] += gb.pow(2).sum().item()
    cosines = {k: v[0] / max((v[1] * v[2]) ** 0.5, 1e-30) for k, v in groups.items()}
    print("[B=32 train] per-stage gradient cosine bf16 vs f32: " + ", ".join(f"{k}={v:.4f}" for k, v in sorted(cosines.items())))
    for k, v in cosines.items():
        floor = 0.97 if k in ("dec_layer1", "dec_layer2", "decoder_other") else (0.95 if k == "dec_layer0" else 0.80)
        assert v > floor, f"{k}: gradient cosine {v:.4f} below {floor}"
    r = b["routes"]
    print(f"[B=32 train] bf16 routes {r}; f32 routes {f['routes']}")
    # the persistent kernels are bf16-only and chosen by size: at this batch the model must reach them (forward products >= 2 GFLOP,
    # the 3x3 convolutions of the fused-MBConv stages and their data gradients, the large dense weight gradients)
    assert r["gemm_big"] >= 4, r
    assert r["gemm_big_conv"] >= 8, r
    assert r["wgrad_big"] >= 2, r
    assert r["gemm_tall"] >= 8, r          # the 1x1 projections of the fused-MBConv stages and their data gradients
    assert r["mbconv_bwd"] >= 20, r
    assert r["bn_pool_se"] + r["mbconv_fwd"] >= 20, r     # the late MBConv blocks' squeeze-and-excite seam in one launch (or inside the block kernel)
    assert f["routes"]["gemm_big"] == 0 and f["routes"]["wgrad_big"] == 0


def test_read_loss_raises_on_a_raised_device_error_word():
    """train_step / read_loss surface the device error word: a raised bit (here set by hand through a loss target outside the vocabulary)
    makes read_loss raise instead of returning a loss computed from skipped elements."""
    import satrn_amd
    torch.manual_seed(21)
    m = bench.make_model("bf16", H, W, 0.1).to("cuda:0")
    m.train()
    img, exp = bench.synth(4, H, W, 16, 3, torch.device("cuda:0"))
    m.train_step(img, exp, 1e-4)
    assert m.read_loss()[0] > 0
    bad = exp.clone()
    bad[0, 3] = 9999
    m.train_step(img, bad, 1e-4)
    with pytest.raises(satrn_amd.SatrnError):
        m.read_loss()
    m.train_step(img, exp, 1e-4)     # the word was cleared by the failing read
    assert m.read_loss()[0] > 0
