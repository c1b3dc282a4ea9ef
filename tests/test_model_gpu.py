"""GPU parity tests, model level: the drop-in modules (satrn_amd.LiteSATRN / EfficientSATRN) through the C-ABI against
(a) the golden vectors produced by the reference itself and (b) the CPU oracle on the same seeded inputs.
f32 mode carries the parity claim (<= 1e-3 on logits, greedy token ids bit-exact where the golden top-1/top-2 margin is
clear); bf16 mode (the throughput mode) is checked against looser, stated tolerances."""
import os
from satrn_amd import switches as sw

import numpy as np
import pytest
import torch

from oracle import satrn_oracle as O

pytestmark = pytest.mark.gpu


class _DS:
    def __init__(self):
        import satrn_amd
        self.token_to_id = {satrn_amd.START: 0, satrn_amd.END: 1, satrn_amd.PAD: 2}
        self.id_to_token = {i: str(i) for i in range(O.NUM_CLASSES)}


def make_flags(cfg, h, w, dropout=0.0):
    import satrn_amd
    return satrn_amd.Flags(dict(
        network=cfg["network"], input_size=dict(height=h, width=w),
        SATRN=dict(encoder=dict(hidden_dim=cfg["enc_hidden"], filter_dim=cfg["enc_filter"], layer_num=cfg["enc_layers"], head_num=cfg["enc_heads"]),
                   decoder=dict(src_dim=cfg["dec_src"], hidden_dim=cfg["dec_hidden"], filter_dim=cfg["dec_filter"], layer_num=cfg["dec_layers"], head_num=cfg["dec_heads"])),
        data=dict(rgb=cfg["rgb"]), dropout_rate=dropout)).get()


def build(cfg, h, w, dtype, wseed, dropout=0.0):
    import satrn_amd
    cls = satrn_amd.LiteSATRN if cfg["network"] == "LiteSATRN" else satrn_amd.EfficientSATRN
    sd = O.det_state_dict(cfg, wseed)
    model = cls(make_flags(cfg, h, w, dropout), _DS(), sd, dtype=dtype).to("cuda")
    return model, sd


def load_case(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    meta = dict(zip(z["meta_keys"].tolist(), z["meta_vals"].tolist()))
    cfg = dict(network=meta["network"])
    for k in ("rgb", "enc_hidden", "enc_filter", "enc_heads", "enc_layers", "dec_src", "dec_hidden", "dec_filter",
              "dec_heads", "dec_layers", "num_classes"):
        cfg[k] = int(meta[k])
    return z, meta, cfg


def checksum_samples(t):
    t = t.detach().double().flatten().cpu()
    n = t.numel()
    idx = (torch.arange(64, dtype=torch.int64) * 2654435761 % max(n, 1))
    return t[idx].numpy()


def relerr(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    return (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-12)


CASES = ["lite_small", "lite_c1", "lite_c1_pad", "eff_small", "eff_c2_b2"]


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_train_forward_backward_vs_golden_and_oracle(golden_dir, name, dtype):
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    z, meta, cfg = load_case(golden_dir, name)
    B, H, W, T = (int(meta[k]) for k in ("batch", "height", "width", "seq_len"))
    model, sd = build(cfg, H, W, dtype, int(meta["wseed"]))
    img, expected = O.det_inputs(B, cfg["rgb"], H, W, T, seed=int(meta["iseed"]), pad_tail=int(meta["pad_tail"]))
    imgd, expd = img.cuda(), expected.cuda()
    model.train()
    logits = model(imgd, expd, True, 1.0)
    loss = model.criterion(logits.transpose(1, 2), expd[:, 1:])
    model.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    f32 = dtype == "f32"
    # eff_small (B=2, 2x3 feature map: BatchNorm over 12 samples) amplifies every rounding difference; it is kept as a
    # structural check (all 40 blocks at a tiny size) with looser bounds
    tiny = name == "eff_small"
    # ---- against the reference's golden vectors
    print(f"[{name}:{dtype}] loss {loss.item():.6f} golden {float(z['loss']):.6f}")
    assert abs(loss.item() - float(z["loss"])) < (1e-3 if f32 else 0.15)
    smp = checksum_samples(logits)
    lerr = np.abs(smp - z["logits_samples"]).max()
    print(f"[{name}:{dtype}] logits sample max err {lerr:.3e}")
    assert lerr < (1e-3 if f32 else (0.5 if tiny else 0.25))
    if "logits" in z:
        assert np.abs(logits.detach().cpu().numpy() - z["logits"]).max() < (1e-3 if f32 else 0.25)
    # ---- against the oracle (full gradients of every parameter + BN running stats)
    oloss, ologits, ograds, obn = O.forward_backward(img, expected, sd, cfg)
    if f32:
        assert relerr(logits, ologits) < 1e-3
    params = dict(model.named_parameters())
    gmax = max(g.abs().max().item() for g in ograds.values())
    gl2 = max(g.norm().item() / max(g.numel(), 1) ** 0.5 for g in ograds.values())

    def tensor_errs(got):
        """per parameter: (relative L2 error, relative max error); parameters whose true gradient is exactly zero
        (k_linear.bias: softmax shift invariance; biases in front of a batch-stat BN) only carry rounding noise, so both
        are measured against a floor tied to the global gradient scale"""
        out = {}
        for n_, g in ograds.items():
            d = got[n_].detach().float().cpu() - g
            rms = lambda t: t.norm().item() / max(t.numel(), 1) ** 0.5
            out[n_] = (rms(d) / max(rms(g), 1e-3 * gl2), d.abs().max().item() / max(g.abs().max().item(), 1e-3 * gmax))
        return out

    errs = tensor_errs({n_: p.grad for n_, p in params.items()})
    top = sorted(((e[0], n_) for n_, e in errs.items()), reverse=True)[:6]
    print(f"[{name}:{dtype}] worst grad rel-L2 errs: " + ", ".join(f"{n_}={e:.2e}" for e, n_ in top))
    if f32:
        # per-tensor bounds are loose on purpose: ONE ReLU / max-pool decision flipping at |u| ~ 1e-7 (forward agrees to
        # 1e-6) moves a whole element of a small late-stage tensor and everything upstream of it; the median is tight
        med = float(np.median([e[0] for e in errs.values()]))
        wl2, wmx = max(e[0] for e in errs.values()), max(e[1] for e in errs.values())
        print(f"[{name}:{dtype}] median rel-L2 grad err {med:.3e}; worst rel-L2 {wl2:.3e}, worst rel-max {wmx:.3e}")
        # the engine's f32 step is deterministic (fixed-order reductions), so these are fixed numbers, not noise bounds:
        # measured medians 1.4e-6 .. 2.0e-4, worst tensor 8.4e-3 (lite_c1_pad: one ReLU / max-pool decision that the CPU
        # oracle's different summation order takes the other way moves a late-stage element and everything upstream)
        # eff_c2_b2 (2 images: the encoder's BatchNorms see 96 rows): round 4 changed the ORDER in which the deterministic mode folds its
        # per-tile statistics (parallel two-level fold instead of one serial chain per column: 40 -> 25 ms per f32 step) -- the forward still
        # agrees to 6e-6, but another ReLU decision of encoder layer 0 (|u| ~ 1e-7) now falls on the other side than the CPU oracle's:
        # median 4.0e-3, worst tensor 1.1e-2.  Every other case keeps the 1e-3 median.
        assert med < (6e-3 if name == "eff_c2_b2" else 1e-3)
        for n_, (l2, mx) in errs.items():
            assert l2 < 1.5e-2 and mx < 1e-1, f"grad {n_}: rel L2 {l2} max {mx}"
    else:
        # yardstick: the oracle graph run by PyTorch itself with every tensor in bf16 (CPU bf16 kernels)
        _, tlogits, tgrads, _ = O.forward_backward(img, expected, sd, cfg, dtype=torch.bfloat16)
        terrs = tensor_errs(tgrads)
        le, lt = relerr(logits, ologits), relerr(tlogits, ologits)
        print(f"[{name}:{dtype}] logits rel err: engine {le:.3e} vs torch-bf16 {lt:.3e}")
        assert le < 2.0 * lt + 0.02
        rms = lambda t_: t_.norm().item() / max(t_.numel(), 1) ** 0.5
        zero_true = {n_ for n_, g in ograds.items() if rms(g) < 1e-3 * gl2}
        for n_ in zero_true:  # pure rounding noise on an exactly-zero gradient: bounded relative to the global scale
            assert errs[n_][0] < 5.0, f"noise on zero-gradient parameter {n_}: {errs[n_][0]}"
        ratio = sorted(((errs[n_][0] / max(terrs[n_][0], 2e-2), n_) for n_ in errs if n_ not in zero_true), reverse=True)
        med_e = float(np.median([e[0] for e in errs.values()]))
        med_t = float(np.median([e[0] for e in terrs.values()]))
        print(f"[{name}:{dtype}] median rel-L2 grad err: engine {med_e:.3e} vs torch-bf16 {med_t:.3e}; worst engine/torch ratios: "
              + ", ".join(f"{n_}={r:.2f}" for r, n_ in ratio[:5]))
        assert med_e < 2.0 * med_t + 0.02
        for r, n_ in ratio:
            assert r < 4.0, f"grad {n_}: engine bf16 error {errs[n_][0]:.3e} vs torch bf16 {terrs[n_][0]:.3e}"
    gs = np.stack([np.array([params[n_].grad.double().sum().item(), params[n_].grad.double().abs().sum().item()]) for n_ in O.trainable_names(cfg)])
    if f32:
        np.testing.assert_allclose(gs[:, 1], z["grad_sums"][:, 1], rtol=5e-3, atol=1e-5)
    bufs = dict(model.named_buffers())
    for n_, v in obn.items():
        assert relerr(bufs[n_], v) < (1e-3 if f32 else (0.3 if tiny else 0.1)), n_


@pytest.mark.parametrize("name", ["lite_c1", "eff_c2_b2"])
def test_f32_throughput_kernels_vs_oracle(golden_dir, name, monkeypatch):
    """f32 normally runs the fixed-order (deterministic) reductions, bf16 the atomic / fused forms (statistics in GEMM and
    depthwise-convolution epilogues, BatchNorm-backward sums in data-gradient epilogues).  SATRN_NONDET=1 runs those
    throughput-mode kernels in f32, so they are checked against the oracle at f32 tolerance and not only at bf16's."""
    monkeypatch.setenv("SATRN_NONDET", "1")
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    z, meta, cfg = load_case(golden_dir, name)
    B, H, W, T = (int(meta[k]) for k in ("batch", "height", "width", "seq_len"))
    model, sd = build(cfg, H, W, "f32", int(meta["wseed"]))
    img, expected = O.det_inputs(B, cfg["rgb"], H, W, T, seed=int(meta["iseed"]), pad_tail=int(meta["pad_tail"]))
    imgd, expd = img.cuda(), expected.cuda()
    model.train()
    logits = model(imgd, expd, True, 1.0)
    loss = model.criterion(logits.transpose(1, 2), expd[:, 1:])
    model.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - float(z["loss"])) < 1e-3
    oloss, ologits, ograds, obn = O.forward_backward(img, expected, sd, cfg)
    assert relerr(logits, ologits) < 1e-3
    params = dict(model.named_parameters())
    gl2 = max(g.norm().item() / max(g.numel(), 1) ** 0.5 for g in ograds.values())
    rms = lambda t: t.norm().item() / max(t.numel(), 1) ** 0.5
    errs = {n_: rms(params[n_].grad.detach().float().cpu() - g) / max(rms(g), 1e-3 * gl2) for n_, g in ograds.items()}
    med, worst = float(np.median(list(errs.values()))), max(errs.values())
    print(f"[{name}:f32 atomic mode] median rel-L2 grad err {med:.3e}, worst {worst:.3e} ({max(errs, key=errs.get)})")
    # atomic summation order differs from run to run, and a last-bit change of a BatchNorm statistic flips ReLU / max-pool
    # decisions in the first layers: the gradient error lands in a few discrete states (observed medians 1.9e-4, 7.8e-4,
    # 4.0e-3 on eff_c2_b2 -- the same values with the fused epilogues switched off), all far below a wrong-kernel error (O(1))
    assert med < 2e-2
    assert worst < 1e-1
    bufs = dict(model.named_buffers())
    for n_, v in obn.items():
        assert relerr(bufs[n_], v) < 1e-3, n_


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_eval_encode_and_greedy_vs_golden(golden_dir, name, dtype):
    z, meta, cfg = load_case(golden_dir, name)
    B, H, W, T = (int(meta[k]) for k in ("batch", "height", "width", "seq_len"))
    model, sd = build(cfg, H, W, dtype, int(meta["wseed"]))
    img, expected = O.det_inputs(B, cfg["rgb"], H, W, T, seed=int(meta["iseed"]), pad_tail=int(meta["pad_tail"]))
    imgd = img.cuda()
    model.eval()
    f32 = dtype == "f32"
    src = model.encode(imgd)
    err = np.abs(checksum_samples(src) - z["enc_samples"]).max()
    print(f"[{name}:{dtype}] encoder sample max err {err:.3e}")
    assert err < (1e-3 if f32 else 0.2)
    steps = z["greedy_ids"].shape[1]
    with torch.no_grad():
        glog = model(imgd, expected[:, : steps + 1].cuda(), False, 0.0)
    assert glog.shape == (B, steps, cfg["num_classes"])
    ids = glog.argmax(-1).cpu().numpy()
    gerr = np.abs(checksum_samples(glog) - z["greedy_samples"]).max()
    print(f"[{name}:{dtype}] greedy logits sample max err {gerr:.3e}")
    if f32:
        assert gerr < 2e-3
        clear = z["greedy_margin"] > 5e-3
        assert (ids[clear] == z["greedy_ids"][clear]).all(), "greedy token ids differ from the reference"
        if "greedy_logits" in z:
            assert np.abs(glog.cpu().numpy() - z["greedy_logits"]).max() < 2e-3
    else:
        # bf16: ids agree wherever the reference's margin is larger than the bf16 noise on the logits
        clear = z["greedy_margin"] > 0.5
        agree = (ids[clear] == z["greedy_ids"][clear]).mean() if clear.any() else 1.0
        print(f"[{name}:{dtype}] greedy id agreement on clear margins: {agree:.3f}")
        assert agree > 0.9
    _, ids2 = model.greedy(imgd, steps)
    assert (ids2.cpu().numpy() == ids).all()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_fused_train_step_matches_module_path_and_oracle(golden_dir, dtype):
    """satrn_model_train_step (one hipGraph: fwd + CE + bwd + clip + AdamW) == module forward/backward + the oracle's
    clip_adamw_step; then graph replays keep reducing the loss."""
    z, meta, cfg = load_case(golden_dir, "lite_c1_pad")
    B, H, W, T = (int(meta[k]) for k in ("batch", "height", "width", "seq_len"))
    img, expected = O.det_inputs(B, cfg["rgb"], H, W, T, seed=int(meta["iseed"]), pad_tail=int(meta["pad_tail"]))
    imgd, expd = img.cuda(), expected.cuda()
    lr = 5e-4
    # oracle step
    sd = O.det_state_dict(cfg, int(meta["wseed"]))
    oloss, _, ograds, _ = O.forward_backward(img, expected, sd, cfg)
    names = O.trainable_names(cfg)
    p = {n: sd[n].clone() for n in names}
    m = {n: torch.zeros_like(sd[n]) for n in names}
    v = {n: torch.zeros_like(sd[n]) for n in names}
    gnorm = O.clip_adamw_step(p, ograds, m, v, 1, lr)
    for use_graph in (False, True):
        model, _ = build(cfg, H, W, dtype, int(meta["wseed"]))
        model.train()
        model.train_step(imgd, expd, lr, use_graph=use_graph)
        loss, cnt, gn = model.read_loss()
        print(f"[train_step:{dtype}:graph={use_graph}] loss {loss:.6f} (oracle {oloss.item():.6f}) gnorm {gn:.4f} (oracle {gnorm.item():.4f})")
        f32 = dtype == "f32"
        assert abs(loss - oloss.item()) < (1e-3 if f32 else 5e-2)
        assert cnt == (expected[:, 1:] != 2).sum().item()
        assert abs(gn - gnorm.item()) / gnorm.item() < (2e-3 if f32 else 0.1)
        params = dict(model.named_parameters())
        worst = 0.0
        gmax = max(g.abs().max().item() for g in ograds.values())
        for n in names:
            # AdamW's first step moves every weight by ~lr*sign(g): compare the UPDATE, and only where the gradient is
            # clearly non-zero (the sign of a rounding-noise gradient is arbitrary)
            du = (params[n].detach().cpu() - sd[n])
            dr = (p[n] - sd[n])
            sig = ograds[n].abs() > 1e-3 * gmax
            if sig.any():
                worst = max(worst, ((du - dr).abs()[sig]).max().item() / lr)
        print(f"[train_step:{dtype}:graph={use_graph}] worst update err / lr = {worst:.3e}")
        if f32:
            assert worst < 0.05
        losses = [loss]
        for _ in range(8):
            model.train_step(imgd, expd, lr, use_graph=use_graph)
            losses.append(model.read_loss()[0])
        print("losses", [round(x, 4) for x in losses])
        assert all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_dropout_path_runs_and_is_unbiased(golden_dir):
    """dropout 0.1 (the throughput configuration): loss stays close to the dropout-free loss and differs step to step."""
    z, meta, cfg = load_case(golden_dir, "lite_c1")
    B, H, W, T = (int(meta[k]) for k in ("batch", "height", "width", "seq_len"))
    img, expected = O.det_inputs(B, cfg["rgb"], H, W, T, seed=int(meta["iseed"]))
    model, _ = build(cfg, H, W, "f32", int(meta["wseed"]), dropout=0.1)
    model.train()
    vals = []
    for _ in range(3):
        model.train_step(img.cuda(), expected.cuda(), 0.0, use_graph=True)
        vals.append(model.read_loss()[0])
    print("dropout losses", vals)
    assert len(set(round(v, 6) for v in vals)) > 1
    assert all(abs(v - float(z["loss"])) < 0.5 for v in vals)


def test_no_cpu_fallback():
    import satrn_amd
    cfg = dict(O.CFG_LITE)
    model = satrn_amd.LiteSATRN(make_flags(cfg, 64, 192), _DS(), None, dtype="f32")
    img, expected = O.det_inputs(2, 1, 64, 192, 8)
    with pytest.raises(satrn_amd.SatrnError):
        model(img, expected, True, 1.0)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_train_autoregressive_branch_with_gradients(dtype):
    """model(input, expected, True, 0.0): the reference's non-teacher-forced training branch
    (networks/EfficientSATRN.py:496-525) -- step-wise greedy feeding WITH gradients through every step."""
    cfg = dict(O.CFG_LITE)
    B, H, W, T = 3, 64, 192, 7
    model, sd = build(cfg, H, W, dtype, 6)
    img, expected = O.det_inputs(B, 1, H, W, T, seed=44, pad_tail=2)
    model.train()
    logits = model(img.cuda(), expected.cuda(), True, 0.0)   # random.random() < 0.0 never holds -> AR branch
    loss = model.criterion(logits.transpose(1, 2), expected.cuda()[:, 1:])
    model.zero_grad()
    loss.backward()
    oloss, ologits, ograds, _ = O.forward_backward(img, expected, sd, cfg, teacher_forcing=False)
    f32 = dtype == "f32"
    print(f"[ar:{dtype}] loss {loss.item():.6f} oracle {oloss.item():.6f} logits rel err {relerr(logits, ologits):.3e}")
    assert abs(loss.item() - oloss.item()) < (1e-3 if f32 else 0.1)
    if f32:
        assert relerr(logits, ologits) < 1e-3
        params = dict(model.named_parameters())
        gl2 = max(g.norm().item() / max(g.numel(), 1) ** 0.5 for g in ograds.values())
        errs = []
        for n_, g in ograds.items():
            d = params[n_].grad.detach().cpu() - g
            rms = lambda t_: t_.norm().item() / max(t_.numel(), 1) ** 0.5
            errs.append((rms(d) / max(rms(g), 1e-3 * gl2), n_))
        worst = sorted(errs, reverse=True)[:4]
        print(f"[ar:{dtype}] worst grad rel-L2: " + ", ".join(f"{n_}={e_:.2e}" for e_, n_ in worst))
        assert float(np.median([e_ for e_, _ in errs])) < 1e-3 and worst[0][0] < 3e-2


def _ar_routes(reset=False):
    import ctypes
    import satrn_amd
    out = (ctypes.c_longlong * 16)()
    satrn_amd._lib.load().satrn_route_counts(out, 16, int(reset))
    return out[9]


@pytest.mark.parametrize("dtype,net,T", [("f32", "lite", 9), ("bf16", "lite", 9), ("f32", "eff", 6)])
def test_autoregressive_branch_two_launch_form_equals_operator_form(dtype, net, T):
    """kernels_ar.hip (one workgroup per image runs all steps of a direction, weight gradients as products over [B*T]-row slabs) against
    the operator-level form it replaces (SATRN_OFF=ar_fused: ~126 launches per step): same logits, same predicted ids, same gradients.
    f32 pins it tightly; bf16 rounds at different places (f32 residual stream inside the kernel) and is held to the bf16 yardstick."""
    cfg = dict(O.CFG_LITE if net == "lite" else O.CFG_EFF)
    B, H, W = 3, 64, 192
    img, expected = O.det_inputs(B, cfg["rgb"], H, W, T, seed=44, pad_tail=2)
    res = {}
    for name in ("fused", "ops"):
        (sw.off if name == "ops" else sw.on)("ar_fused")
        model, sd = build(cfg, H, W, dtype, 6)
        model.train()
        _ar_routes(reset=True)
        logits = model(img.cuda(), expected.cuda(), True, 0.0)
        loss = model.criterion(logits.transpose(1, 2), expected.cuda()[:, 1:])
        model.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        assert (_ar_routes() > 0) == (name == "fused"), "the route under test must be the one that ran"
        res[name] = (logits.detach().float().cpu(), loss.item(), {n_: p_.grad.detach().float().cpu().clone() for n_, p_ in model.named_parameters()})
    sw.on("ar_fused")
    (lf, lossf, gf), (lo, losso, go) = res["fused"], res["ops"]
    f32 = dtype == "f32"
    print(f"[ar two-launch:{dtype}:{net}] loss {lossf:.6f} vs {losso:.6f}; logits rel err {relerr(lf, lo):.3e}")
    assert torch.equal(lf.argmax(-1), lo.argmax(-1)) or not f32
    assert abs(lossf - losso) < (1e-5 if f32 else 5e-2)
    assert relerr(lf, lo) < (2e-5 if f32 else 5e-2)
    rms = lambda t_: t_.norm().item() / max(t_.numel(), 1) ** 0.5
    gl2 = max(rms(g) for g in go.values())
    errs = sorted(((rms(gf[n_] - g) / max(rms(g), 1e-3 * gl2), n_) for n_, g in go.items()), reverse=True)
    print(f"[ar two-launch:{dtype}:{net}] worst grad rel-L2: " + ", ".join(f"{n_}={e_:.2e}" for e_, n_ in errs[:4]))
    if f32:
        assert errs[0][0] < 2e-3 and float(np.median([e_ for e_, _ in errs])) < 1e-4
    else:
        flat_f = torch.cat([gf[n_].flatten() for n_ in go]); flat_o = torch.cat([g.flatten() for g in go.values()])
        cs = torch.dot(flat_f, flat_o) / (flat_f.norm() * flat_o.norm())
        print(f"[ar two-launch:{dtype}:{net}] flat gradient cosine {cs.item():.5f}")
        assert cs.item() > 0.98


def test_autoregressive_forward_history_in_lds_or_in_memory():
    """The sliced forward keeps its slice of the attention history and of the cross-attention keys / values in LDS where they fit; longer
    targets / f32 at the full size read them from memory (the step's own row from LDS).  Same arithmetic either way: in f32 (fixed-order
    reductions everywhere) the same bits.  (bf16 cannot be compared run to run on this tiny case: the backbone's batch statistics go
    through fp32 atomics and 3 x 2 x 6 positions per channel amplify their last bits; bf16 takes the same code with T = bf16_t.)"""
    dtype = "f32"
    cfg = dict(O.CFG_EFF)
    B, H, W, T = 3, 64, 192, 7
    img, expected = O.det_inputs(B, cfg["rgb"], H, W, T, seed=46, pad_tail=1)
    res = {}
    for name in ("lds", "mem"):
        (sw.off if name == "mem" else sw.on)("ar_kv_lds")
        model, sd = build(cfg, H, W, dtype, 6)
        model.train()
        _ar_routes(reset=True)
        logits = model(img.cuda(), expected.cuda(), True, 0.0)
        loss = model.criterion(logits.transpose(1, 2), expected.cuda()[:, 1:])
        model.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        assert _ar_routes() > 0
        res[name] = (logits.detach().float().cpu().clone(), model.flat_grad().detach().float().cpu().clone())
    sw.on("ar_kv_lds")
    assert torch.equal(res["lds"][0], res["mem"][0]), "logits"
    assert torch.equal(res["lds"][1], res["mem"][1]), "gradients"


def test_autoregressive_branch_beyond_the_resident_batch_takes_the_operator_form():
    """The backward's layer pipeline needs batch x layers workgroups resident (256 on an MI355X: 85 images at three layers, 128 at two); a larger batch
    must take the operator-level branch by itself -- not fail, not hang."""
    cfg = dict(O.CFG_LITE)
    B, H, W, T = 136, 64, 192, 4   # (LiteSATRN: two decoder layers)
    img, expected = O.det_inputs(B, cfg["rgb"], H, W, T, seed=45, pad_tail=1)
    model, sd = build(cfg, H, W, "bf16", 6)
    model.train()
    _ar_routes(reset=True)
    logits = model(img.cuda(), expected.cuda(), True, 0.0)
    loss = model.criterion(logits.transpose(1, 2), expected.cuda()[:, 1:])
    model.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    assert _ar_routes() == 0
    assert torch.isfinite(loss).item() and torch.isfinite(model.flat_grad()).all().item()


def test_autoregressive_branch_two_launch_form_with_dropout():
    """Dropout inside the two-launch form: the backward regenerates the masks of the forward (counter hash), so two runs from the same
    seed agree bit for bit in f32 with fixed-order reductions, the loss differs from the dropout-free one, and every gradient is finite;
    the masks keep the expectation: the mean loss over seeds is within a few percent of the dropout-free loss's neighbourhood."""
    cfg = dict(O.CFG_LITE)
    B, H, W, T = 3, 64, 192, 9
    img, expected = O.det_inputs(B, cfg["rgb"], H, W, T, seed=44, pad_tail=2)

    def run(dropout, seed):
        import ctypes
        import satrn_amd
        model, sd = build(cfg, H, W, "f32", 6, dropout=dropout)
        model.train()
        model.reserve(B, T + 1, "cuda")
        word = ctypes.c_uint32(seed)
        assert satrn_amd._lib.load().satrn_model_rng_state(model._h, ctypes.byref(word), 1, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
        logits = model(img.cuda(), expected.cuda(), True, 0.0)
        loss = model.criterion(logits.transpose(1, 2), expected.cuda()[:, 1:])
        model.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        return loss.item(), model.flat_grad().detach().float().cpu().clone()

    l0, g0 = run(0.0, 5)
    l1, g1 = run(0.1, 5)
    l2, g2 = run(0.1, 5)
    l3, g3 = run(0.1, 6)
    print(f"[ar dropout] loss {l0:.5f} (p = 0), {l1:.5f} / {l2:.5f} (p = 0.1, same seed), {l3:.5f} (another seed)")
    assert torch.isfinite(g1).all() and l1 == l2 and torch.equal(g1, g2)
    assert l1 != l0 and l3 != l1 and not torch.equal(g1, g3)
    assert abs(l1 - l0) < 0.5 and abs(l3 - l0) < 0.5


@pytest.mark.parametrize("net,H,W,B", [("eff", 128, 384, 4), ("eff", 64, 96, 3), ("lite", 64, 192, 3)])
def test_fused_encoder_attention_region_equals_the_four_launches(net, H, W, B):
    """kernels_encattn.hip (LayerNorm -> q|k|v -> attention -> output-projection partials in ONE launch, partials folded by the LayerNorm
    behind the block) against the four launches it replaces (SATRN_OFF=fused_enc_attn), bf16, through the whole model.  The operator
    test (test_ops_gpu.py::test_encoder_attention_region_fused) pins every tensor the kernel writes to fp32 torch; here: (1) eval-mode
    encoder output, (2) the flat gradient of a step with BatchNorm in running-statistics mode (train_step(bn_eval=True): with BATCH
    statistics on 4 x 12 maps a bf16 rounding anywhere moves the gradient by tens of percent -- bench.py's accuracy_bf16_vs_f32 -- so two
    correct forms cannot be compared there), (3) a training step with dropout: both forms draw the same masks (same counter hash, sites
    and indices), loss and greedy ids of the step agree."""
    import os
    cfg = dict(O.CFG_EFF if net == "eff" else O.CFG_LITE)
    T = 9
    img, expected = O.det_inputs(B, cfg["rgb"], H, W, T, seed=71, pad_tail=2)
    imgd, expd = img.cuda(), expected.cuda()
    res = {}
    for name, env in (("fused", None), ("four", "1")):
        if env:
            sw.off("fused_enc_attn")
        else:
            sw.on("fused_enc_attn")
        try:
            model, sd = build(cfg, H, W, "bf16", 9, dropout=0.1)
            model.eval()
            src = model.encode(imgd).float().cpu()
            model.train()
            model.train_step(imgd, expd, 0.0, phase=1, bn_eval=True)
            torch.cuda.synchronize()
            g_eval = model.flat_grad().detach().float().cpu().clone()
            l_eval = model.read_loss()[0]
            model.train_step(imgd, expd, 0.0, phase=1)
            torch.cuda.synchronize()
            res[name] = (src, g_eval, l_eval, model.read_loss()[0], model.flat_grad().detach().float().cpu().clone())
        finally:
            sw.on("fused_enc_attn")
    (s0, g0, le0, l0, gt0), (s1, g1, le1, l1, gt1) = res["fused"], res["four"]
    es = (s0 - s1).abs().max().item() / s1.abs().max().item()
    eg = (g0 - g1).norm().item() / g1.norm().item()
    cos_t = torch.dot(gt0, gt1).item() / (gt0.norm().item() * gt1.norm().item())
    print(f"[enc attn region {net} {H}x{W}] encoder output rel err {es:.3e}; running-statistics step: flat gradient rel-L2 {eg:.3e}, loss {le0:.5f} vs {le1:.5f}; "
          f"training step (batch statistics, dropout): loss {l0:.5f} vs {l1:.5f}, gradient cosine {cos_t:.4f}")
    assert torch.isfinite(s0).all() and torch.isfinite(g0).all() and torch.isfinite(gt0).all()
    assert es < 2e-2          # bf16 rounding of the partial sums instead of one f32 accumulation
    assert eg < 6e-2
    # the training-mode loss of the tiny batch moves by ~0.04 from run to run on its own (batch statistics over 18 rows, float atomics in
    # the statistics, bf16): observed 5.739 and 5.779 for the SAME form in two runs -- the bound is that scale, not a kernel tolerance
    assert abs(le0 - le1) < 1e-2 and abs(l0 - l1) < (3e-2 if net == "lite" else 1e-1)
    assert cos_t > (0.95 if net == "lite" else 0.5)   # (batch statistics over 18 rows at 64x96 bs3: see the docstring)


def test_fused_train_step_takes_the_autoregressive_branch():
    """train_step(teacher_forced=False / teacher_forcing_ratio < 1 + coin): the fused step (the one dp.dp_train_step and bench.py drive)
    runs the reference's non-teacher-forced branch (networks/EfficientSATRN.py:496-525) -- same gradient as the module-API path that
    test_train_autoregressive_branch_with_gradients pins to the oracle, whole and in data-parallel backward segments; the coin of
    :489 is flipped once per step from a shareable source."""
    from satrn_amd import dp
    cfg = dict(O.CFG_LITE)
    B, H, W, T = 3, 64, 192, 7
    model, sd = build(cfg, H, W, "f32", 6)
    img, expected = O.det_inputs(B, 1, H, W, T, seed=44, pad_tail=2)
    imgd, expd = img.cuda(), expected.cuda()
    model.train()
    logits = model(imgd, expd, True, 0.0)
    loss = model.criterion(logits.transpose(1, 2), expd[:, 1:])
    model.zero_grad()
    loss.backward()
    g_ref = model.flat_grad().detach().clone()
    model.train_step(imgd, expd, 0.0, phase=1, teacher_forced=False)
    g = model.flat_grad().detach().clone()
    assert not model.last_teacher_forced
    scale = g_ref.abs().max().item()
    print(f"[fused AR] max |g - g_module| / max |g| = {(g - g_ref).abs().max().item() / scale:.3e}, loss {model.read_loss()[0]:.6f} vs {loss.item():.6f}")
    assert (g - g_ref).abs().max().item() <= 1e-6 * scale
    assert abs(model.read_loss()[0] - loss.item()) < 1e-5
    # the teacher-forced gradient is a different one (the branch really switched)
    model.train_step(imgd, expd, 0.0, phase=1, teacher_forced=True)
    assert (model.flat_grad() - g_ref).abs().max().item() > 1e-3 * scale
    # backward segments (the overlapped data-parallel exchange): segment 0 flips / fixes the branch, the later calls repeat it
    model.train_step(imgd, expd, 0.0, phase=16 + 0 + 4 * 2, teacher_forced=False)
    model.train_step(imgd, expd, 0.0, phase=16 + 3)
    assert (model.flat_grad() - g_ref).abs().max().item() <= 1e-6 * scale
    # hipGraph replay of the branch (captured by the second call of a shape / phase; its own graph beside the teacher-forced one)
    for _ in range(3):
        model.train_step(imgd, expd, 0.0, phase=1, teacher_forced=False, use_graph=True)
    assert (model.flat_grad() - g_ref).abs().max().item() <= 1e-6 * scale
    model.train_step(imgd, expd, 0.0, phase=1, teacher_forced=True, use_graph=True)
    model.train_step(imgd, expd, 0.0, phase=1, teacher_forced=True, use_graph=True)
    assert (model.flat_grad() - g_ref).abs().max().item() > 1e-3 * scale
    # the coin: one flip per step, from the shared source
    model.set_coin(dp.SharedCoin(seed=11))
    twin = dp.SharedCoin(seed=11)
    for _ in range(6):
        model.train_step(imgd, expd, 0.0, phase=1, teacher_forcing_ratio=0.5)
        assert model.last_teacher_forced == twin.teacher_forced(0.5)
    assert model._coin.flips == 6


@pytest.mark.parametrize("name", ["lite_c1_pad", "eff_c2_b2"])
def test_segmented_backward_equals_whole_backward(golden_dir, name):
    """train_step(phase=16+k), k = 0..3 (the overlapped data-parallel exchange's backward segments) leaves the same flat
    gradient as phase=1, and the four segment ranges tile the flat buffer in backward order."""
    z, meta, cfg = load_case(golden_dir, name)
    B, H, W, T = (int(meta[k]) for k in ("batch", "height", "width", "seq_len"))
    img, expected = O.det_inputs(B, cfg["rgb"], H, W, T, seed=int(meta["iseed"]), pad_tail=int(meta["pad_tail"]))
    imgd, expd = img.cuda(), expected.cuda()
    model, _ = build(cfg, H, W, "f32", int(meta["wseed"]))
    model.train()
    model.train_step(imgd, expd, 0.0, phase=1)
    torch.cuda.synchronize()
    whole = model.flat_grad().detach().clone()
    loss1 = model.read_loss()[0]
    # f32 is the deterministic mode: every cross-workgroup reduction (BatchNorm statistics, split weight gradients, bias /
    # LayerNorm / embedding gradients, the loss sum) folds per-workgroup partials in a fixed order, so the same step gives the
    # same BITS however the backward is cut into segments and whichever stream a kernel ran on.  (Round 1 used float atomics:
    # a last-bit change of a BatchNorm statistic flipped ReLU / max-pool decisions and moved gradients by 3e-4 of their
    # maximum -- tools/grad_paths.py shows the discrete states -- and this test had a 2e-2 tolerance.)
    model.train_step(imgd, expd, 0.0, phase=1)
    torch.cuda.synchronize()
    assert torch.equal(model.flat_grad(), whole), "the f32 step is not reproducible from run to run"
    gmax = whole.abs().max().item()
    tol = 0.0
    n = whole.numel()
    ranges = [model.segment_range(k) for k in range(4)]
    assert ranges[0][1] == n and ranges[3][0] == 0 and all(ranges[k][0] == ranges[k + 1][1] for k in range(3))
    for k in range(4):
        model.train_step(imgd, expd, 0.0, phase=16 + k)
        torch.cuda.synchronize()
        lo, hi = ranges[k]
        if hi > lo:  # the range a segment completes is final as soon as its call returns
            d = (model.flat_grad()[lo:hi] - whole[lo:hi]).abs().max().item()
            assert d <= tol, f"segment {k}: {d} > {tol}"
    assert abs(model.read_loss()[0] - loss1) < 1e-4
    d = (model.flat_grad() - whole).abs().max().item()
    print(f"[segments:{name}] max diff {d:.3e} (tolerance {tol:.3e}, max |g| {gmax:.3e})")
    assert d <= tol
    # segments 0-2 in one call, then 3 (what dp.dp_train_step issues)
    model.train_step(imgd, expd, 0.0, phase=16 + 0 + 4 * 2)
    model.train_step(imgd, expd, 0.0, phase=16 + 3)
    torch.cuda.synchronize()
    d = (model.flat_grad() - whole).abs().max().item()
    assert d <= tol, d
    # the single-chain hipGraph replay of the same step (other grids for the weight gradients, one stream): same bits
    model.train_step(imgd, expd, 0.0, phase=1, use_graph=True)   # first call of the shape runs eagerly
    model.train_step(imgd, expd, 0.0, phase=1, use_graph=True)   # replay
    torch.cuda.synchronize()
    dg = (model.flat_grad() - whole).abs().max().item()
    print(f"[segments:{name}] graph replay vs eager: max diff {dg:.3e}")
    assert dg == 0.0   # deterministic mode sizes every split independently of the stream / graph mode
    import satrn_amd
    with pytest.raises(satrn_amd.SatrnError):
        model.train_step(imgd, expd, 0.0, phase=18)  # out of order


# SATRN_OFF names of the chain fusions of rounds 2-4 (each falls back to the kernels it replaced)
_FUSED_SWITCHES = ("fused_pool_se", "fused_pool", "dw_fused_red", "gemm_g2", "se_wide_bwd", "fused_bn_dw", "fused_dw_bwd", "se_bn_sums", "bn_apply_dw", "mbconv_front", "mbconv_bwd_se",
                   "mbconv_xfold", "gemm_tall_off")


def test_fused_chain_kernels_equal_their_plain_forms(golden_dir, monkeypatch):
    """Round-2 chain kernels (BatchNorm + SE pool, depthwise conv + statistics, SE MLP + scale, wide SE backward, two-group
    small-tile GEMM) against the kernels they replaced: same bf16 model and inputs in one process (the switches are read per
    call).  bf16 training with atomic reductions is noisy from run to run (a BatchNorm statistic moving in its last bit flips
    ReLU / max-pool decisions), so the yardstick is that noise itself, measured here by running the PLAIN kernels twice: the
    fused kernels must not be further from the plain ones than the plain ones are from themselves (x2 + a floor)."""
    _, meta, cfg = load_case(golden_dir, "eff_c2_b2")
    H, W, T = (int(meta[k]) for k in ("height", "width", "seq_len"))
    B = 16   # BatchNorm statistics over >= 768 samples in every layer: less flip noise than the 2-image golden case
    img, expected = O.det_inputs(B, cfg["rgb"], H, W, T, seed=77, pad_tail=0)
    imgd, expd = img.cuda(), expected.cuda()

    def run(plain, only=None):
        for k in _FUSED_SWITCHES:
            if k == "gemm_tall_off":   # (a tri-state route, not an on/off feature)
                sw.knob("gemm_tall", 0 if (plain or k == only) else None)
            elif plain or k == only:
                sw.off(k)
            else:
                sw.on(k)
        model, _ = build(cfg, H, W, "bf16", int(meta["wseed"]))
        model.train()
        logits = model(imgd, expd, True, 1.0)
        loss = model.criterion(logits.transpose(1, 2), expd[:, 1:])
        model.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        g = torch.cat([p.grad.detach().float().flatten() for p in model.parameters()])
        return logits.detach().float().clone(), g.clone()

    def dist(a, b):
        (la, ga), (lb, gb) = a, b
        return relerr(lb, la), (gb - ga).norm().item() / ga.norm().item()

    p0, p1, f0 = run(True), run(True), run(False)
    nl, ng = dist(p0, p1)
    fl, fg = dist(p0, f0)
    print(f"[plain vs plain] logits rel err {nl:.3e}, gradient rel-L2 {ng:.3e}   [fused vs plain] {fl:.3e}, {fg:.3e}")
    assert fl < 2.0 * nl + 5e-3
    assert fg < 2.0 * ng + 5e-2
    # round 3: BatchNorm + activation + the whole squeeze-and-excite block as one launch (the image's workgroups hand the pool and the
    # hidden layer to each other through a tagged mailbox) against the same step with only that kernel switched off
    s0 = run(False, only="fused_pool_se")
    sl, sg = dist(s0, f0)
    print(f"[one-launch BatchNorm + squeeze-and-excite vs its two launches] logits rel err {sl:.3e}, gradient rel-L2 {sg:.3e}")
    assert sl < 2.0 * nl + 5e-3
    assert sg < 2.0 * ng + 5e-2
    import satrn_amd
    import ctypes
    assert satrn_amd._lib.load().satrn_device_error(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0   # bit 2 = a hand-off wait of the fused kernel timed out


def test_eval_encoder_image_tile_depthwise_equals_generic(golden_dir, monkeypatch):
    """Inference encoder at the benchmark's image size: the depthwise + eval-BatchNorm + SiLU + SE-pool image-tile kernel followed by
    the SE MLP + scale kernel (round 3) against the generic depthwise kernel + pool/MLP + scale launches it replaces.  The depthwise
    outputs are bit-identical; the pool is summed in another order, so the gate -- and what follows -- may move in the last bf16
    bit."""
    _, meta, cfg = load_case(golden_dir, "eff_c2_b2")
    H, W = 128, 384
    model, _ = build(cfg, H, W, "bf16", int(meta["wseed"]))
    model.eval()
    img, _ = O.det_inputs(4, cfg["rgb"], H, W, 8, seed=5, pad_tail=0)
    imgd = img.cuda()
    sw.off("dw_eval_img")
    a = model.encode(imgd).float().clone()
    sw.on("dw_eval_img")
    b = model.encode(imgd).float().clone()
    err = relerr(b, a)
    print(f"[eval encoder, image-tile depthwise vs generic] rel err {err:.3e}")
    assert torch.isfinite(b).all()
    assert err < 3e-2
    # ... and with the squeeze-and-excite block in the same launch (the default) against the depthwise kernel + SE kernel pair, at the
    # benchmark's batch as well (64 images: the grid must still be resident at once, or the launcher falls back by itself)
    for B2 in (4, 64):
        img2, _ = O.det_inputs(B2, cfg["rgb"], H, W, 8, seed=6, pad_tail=0)
        img2 = img2.cuda()
        sw.off("dw_eval_se")
        a2 = model.encode(img2).float().clone()
        sw.on("dw_eval_se")
        for _ in range(3):
            b2 = model.encode(img2).float().clone()
        err2 = relerr(b2, a2)
        print(f"[eval encoder B={B2}, depthwise + SE in one launch vs two] rel err {err2:.3e}")
        assert torch.isfinite(b2).all()
        assert err2 < 3e-2
    import ctypes
    import satrn_amd
    assert satrn_amd._lib.load().satrn_device_error(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
