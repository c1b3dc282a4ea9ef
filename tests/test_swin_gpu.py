"""GPU parity tests of the SwinTRN path (BASELINE configs[3]): satrn_amd.SWIN through the C-ABI against (a) the vectors the
reference's own SwinTransformer + TransformerDecoder produced (tests/golden/swin_*.npz) and (b) the CPU oracle on the same
seeded inputs.  f32 carries the parity claim (logits <= 1e-3, greedy ids exact where the margin is clear); bf16 is checked
against stated, looser bounds."""
import os
from satrn_amd import switches as sw

import numpy as np
import pytest
import torch

from oracle import satrn_oracle as O
from oracle import swin_oracle as SO

pytestmark = pytest.mark.gpu

CASES = {"swin_tiny": (SO.SWIN_TINY, SO.DEC_TINY), "swin_mid": (SO.SWIN_MID, SO.DEC_MID), "swin_b384_b2": (SO.SWIN_B384, SO.DEC_YAML)}


class _DS:
    def __init__(self):
        import satrn_amd
        self.token_to_id = {satrn_amd.START: 0, satrn_amd.END: 1, satrn_amd.PAD: 2}
        self.id_to_token = {i: str(i) for i in range(O.NUM_CLASSES)}


def build(scfg, dcfg, dtype, wseed, drop_path=0.0, dropout=0.0):
    import satrn_amd
    flags = satrn_amd.Flags(dict(network="SWIN", input_size=dict(height=scfg["img_size"], width=scfg["img_size"]),
                                 SATRN=dict(encoder=dict(hidden_dim=300, filter_dim=600, layer_num=6, head_num=8),   # unused, as in SWIN.yaml
                                            decoder=dict(src_dim=dcfg["dec_src"], hidden_dim=dcfg["dec_hidden"], filter_dim=dcfg["dec_filter"],
                                                         layer_num=dcfg["dec_layers"], head_num=dcfg["dec_heads"])),
                                 data=dict(rgb=3), dropout_rate=dropout)).get()
    sd = SO.det_state_dict(scfg, dcfg, wseed)
    geo = dict(embed_dim=scfg["embed_dim"], depths=scfg["depths"], num_heads=scfg["num_heads"], window_size=scfg["window_size"],
               patch_size=scfg["patch_size"], drop_path_rate=drop_path, head_classes=scfg["head_classes"])
    model = satrn_amd.SWIN(flags, _DS(), sd, dtype=dtype, swin=geo).to("cuda")
    return model, sd


def samples(t):
    t = t.detach().double().flatten().cpu()
    idx = (torch.arange(64, dtype=torch.int64) * 2654435761 % max(t.numel(), 1))
    return t[idx].numpy()


def relerr(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    meta = dict(zip(z["meta_keys"].tolist(), z["meta_vals"].tolist()))
    return z, meta


def test_state_dict_keys_are_the_reference_modules():
    model, sd = build(SO.SWIN_TINY, SO.DEC_TINY, "f32", 1)
    mine = model.state_dict()
    assert list(sorted(mine.keys())) == list(sorted(sd.keys()))
    for k, v in sd.items():
        assert tuple(mine[k].shape) == tuple(v.shape), k
    # the buffers the host mirror fills (relative_position_index, attn_mask) are the reference's formulas
    import satrn_amd
    flags = satrn_amd.Flags(dict(network="SWIN", input_size=dict(height=96, width=96),
                                 SATRN=dict(encoder=dict(hidden_dim=300, filter_dim=600, layer_num=6, head_num=8),
                                            decoder=dict(src_dim=256, hidden_dim=64, filter_dim=64, layer_num=2, head_num=4)),
                                 data=dict(rgb=3), dropout_rate=0.0)).get()
    g = SO.SWIN_TINY
    fresh = satrn_amd.SWIN(flags, _DS(), True, swin=dict(embed_dim=g["embed_dim"], depths=g["depths"], num_heads=g["num_heads"],
                                                           window_size=g["window_size"], head_classes=g["head_classes"]))   # no checkpoint: own init
    for k in ("encoder.layers.0.blocks.1.attn_mask", "encoder.layers.1.blocks.1.attn_mask", "encoder.layers.0.blocks.0.attn.relative_position_index",
              "encoder.layers.3.blocks.0.attn.relative_position_index"):
        assert torch.equal(fresh.state_dict()[k].cpu(), sd[k]), k


@pytest.mark.parametrize("name", ["swin_tiny", "swin_mid", "swin_b384_b2"])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_swin_train_forward_backward_vs_golden_and_oracle(golden_dir, name, dtype):
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    scfg, dcfg = CASES[name]
    z, meta = load(golden_dir, name)
    B, T = int(meta["batch"]), int(meta["seq_len"])
    model, sd = build(scfg, dcfg, dtype, int(meta["wseed"]))
    img, expected = O.det_inputs(B, 3, scfg["img_size"], scfg["img_size"], T, seed=int(meta["iseed"]), pad_tail=int(meta["pad_tail"]))
    imgd, expd = img.cuda(), expected.cuda()
    model.train()
    logits = model(imgd, expd, True, 1.0)
    loss = model.criterion(logits.transpose(1, 2), expd[:, 1:])
    model.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    f32 = dtype == "f32"
    print(f"[{name}:{dtype}] loss {loss.item():.6f} golden {float(z['loss']):.6f}")
    assert abs(loss.item() - float(z["loss"])) < (1e-3 if f32 else 0.1)
    lerr = np.abs(samples(logits) - z["logits_samples"]).max()
    print(f"[{name}:{dtype}] logits sample max err {lerr:.3e}")
    assert lerr < (1e-3 if f32 else 0.3)
    oloss, ologits, ograds, osrc = SO.forward_backward(img, expected, sd, scfg, dcfg)
    if f32:
        assert relerr(logits, ologits) < 1e-3
    params = dict(model.named_parameters())
    gl2 = max(g.norm().item() / max(g.numel(), 1) ** 0.5 for g in ograds.values())
    rms = lambda t: t.norm().item() / max(t.numel(), 1) ** 0.5
    errs = {}
    for n_, g in ograds.items():
        d = params[n_].grad.detach().float().cpu() - g
        errs[n_] = rms(d) / max(rms(g), 1e-3 * gl2)
    top = sorted(((e, n_) for n_, e in errs.items()), reverse=True)[:5]
    med = float(np.median(list(errs.values())))
    print(f"[{name}:{dtype}] median rel-L2 grad err {med:.3e}; worst: " + ", ".join(f"{n_}={e:.2e}" for e, n_ in top))
    if f32:
        # no data-dependent branches except the decoder's ReLUs: f32 gradients agree tightly
        assert med < 1e-4 and top[0][0] < 5e-3
    else:
        assert med < 0.1 and top[0][0] < 1.0
    if f32:
        gs = np.stack([np.array([params[n_].grad.double().sum().item(), params[n_].grad.double().abs().sum().item()]) for n_ in SO.trainable_names(scfg, dcfg)])
        np.testing.assert_allclose(gs[:, 1], z["grad_sums"][:, 1], rtol=5e-3, atol=1e-5)


@pytest.mark.parametrize("name", ["swin_tiny", "swin_mid", "swin_b384_b2"])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_swin_encode_and_greedy_vs_golden(golden_dir, name, dtype):
    scfg, dcfg = CASES[name]
    z, meta = load(golden_dir, name)
    B, T = int(meta["batch"]), int(meta["seq_len"])
    model, sd = build(scfg, dcfg, dtype, int(meta["wseed"]), drop_path=0.5)   # eval: stochastic depth is the identity
    img, _ = O.det_inputs(B, 3, scfg["img_size"], scfg["img_size"], T, seed=int(meta["iseed"]), pad_tail=int(meta["pad_tail"]))
    model.eval()
    f32 = dtype == "f32"
    src = model.encode(img.cuda())
    n = (scfg["img_size"] // scfg["patch_size"] // 8) ** 2
    assert src.shape == (B, n, 8 * scfg["embed_dim"])
    eerr = np.abs(samples(src) - z["enc_samples"]).max()
    print(f"[{name}:{dtype}] encoder sample max err {eerr:.3e}")
    assert eerr < (1e-3 if f32 else 0.15)
    steps = int(meta["greedy_steps"])
    glog, ids = model.greedy(img.cuda(), steps)
    assert glog.shape == (B, steps, O.NUM_CLASSES)
    if f32:
        clear = z["greedy_margin"] > 1e-3
        assert (ids.cpu().numpy()[clear] == z["greedy_ids"][clear]).all()
        assert np.abs(samples(glog) - z["greedy_samples"]).max() < 1e-3
    else:
        clear = z["greedy_margin"] > 0.5
        assert (ids.cpu().numpy()[clear] == z["greedy_ids"][clear]).mean() > 0.9 if clear.any() else True


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_gelu_in_product_epilogue_equals_separate_passes(golden_dir, dtype, monkeypatch):
    """MLP of the Swin block: fc1 with GELU in its epilogue (pre-activation kept beside it) and GELU' applied by fc2's data-gradient
    epilogue, against fc1 -> GELU pass -> fc2 with a separate backward pass.  Forward values are identical by construction (the
    activation is taken of the stored, rounded pre-activation); the backward differs by one rounding of the incoming gradient."""
    scfg, dcfg = CASES["swin_mid"]
    z, meta = load(golden_dir, "swin_mid")
    B, T = int(meta["batch"]), int(meta["seq_len"])
    img, expected = O.det_inputs(B, 3, scfg["img_size"], scfg["img_size"], T, seed=5, pad_tail=0)
    imgd, expd = img.cuda(), expected.cuda()

    def run(passes):
        if passes:
            sw.off("swin_gelu_epilogue")
        else:
            sw.on("swin_gelu_epilogue")
        model, _ = build(scfg, dcfg, dtype, int(meta["wseed"]))
        model.train()
        logits = model(imgd, expd, True, 1.0)
        loss = model.criterion(logits.transpose(1, 2), expd[:, 1:])
        model.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        return logits.detach().float().clone(), torch.cat([p_.grad.detach().float().flatten() for p_ in model.parameters()]).clone()

    (l0, g0), (l1, g1) = run(True), run(False)
    le, ge = relerr(l1, l0), (g1 - g0).norm().item() / g0.norm().item()
    print(f"[gelu epilogue vs passes:{dtype}] logits rel err {le:.3e}, gradient rel-L2 {ge:.3e}")
    assert le < (1e-6 if dtype == "f32" else 2e-2)
    assert ge < (1e-5 if dtype == "f32" else 5e-2)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_window_order_folded_into_layernorm_and_residual(golden_dir, dtype, monkeypatch):
    """shift + window partition written by the LayerNorm in front of the attention and read back by the residual add behind it (RowMap)
    against the four separate permutation passes per block: a pure re-indexing, so the logits are identical and the gradients differ
    only by the order of the float atomics of the weight gradients"""
    scfg, dcfg = CASES["swin_mid"]
    z, meta = load(golden_dir, "swin_mid")
    B, T = int(meta["batch"]), int(meta["seq_len"])
    img, expected = O.det_inputs(B, 3, scfg["img_size"], scfg["img_size"], T, seed=5, pad_tail=0)
    imgd, expd = img.cuda(), expected.cuda()

    def run(passes):
        if passes:
            sw.off("swin_rowmap")
        else:
            sw.on("swin_rowmap")
        model, _ = build(scfg, dcfg, dtype, int(meta["wseed"]))
        model.train()
        logits = model(imgd, expd, True, 1.0)
        loss = model.criterion(logits.transpose(1, 2), expd[:, 1:])
        model.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        return logits.detach().float().clone(), torch.cat([p_.grad.detach().float().flatten() for p_ in model.parameters()]).clone()

    (l0, g0), (l1, g1) = run(True), run(False)
    ge = (g1 - g0).norm().item() / g0.norm().item()
    print(f"[window order folded:{dtype}] logits equal {torch.equal(l0, l1)}, gradient rel-L2 {ge:.3e}")
    assert torch.equal(l0, l1)
    assert ge < (1e-6 if dtype == "f32" else 1e-3)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("drop_path", [0.0, 0.5])
def test_residual_adds_folded_into_layernorm(golden_dir, dtype, drop_path, monkeypatch):
    """x = shortcut + DropPath(branch) folded into the LayerNorm behind it (the attention branch into norm2, the MLP branch into the next
    block's norm1; the backward hands d(sum) to the shortcut and, scaled / window-ordered, to the branch) against the separate residual
    adds: same stochastic-depth masks (same sites), logits identical, gradients equal up to one bf16 rounding of the summed gradient"""
    scfg, dcfg = CASES["swin_mid"]
    z, meta = load(golden_dir, "swin_mid")
    B, T = int(meta["batch"]), int(meta["seq_len"])
    img, expected = O.det_inputs(B, 3, scfg["img_size"], scfg["img_size"], T, seed=5, pad_tail=0)
    imgd, expd = img.cuda(), expected.cuda()

    def run(separate):
        if separate:
            sw.off("swin_add_ln")
        else:
            sw.on("swin_add_ln")
        model, _ = build(scfg, dcfg, dtype, int(meta["wseed"]), drop_path=drop_path)
        model.train()
        logits = model(imgd, expd, True, 1.0)
        loss = model.criterion(logits.transpose(1, 2), expd[:, 1:])
        model.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        return logits.detach().float().clone(), torch.cat([p_.grad.detach().float().flatten() for p_ in model.parameters()]).clone()

    (l0, g0), (l1, g1) = run(True), run(False)
    le, ge = relerr(l1, l0), (g1 - g0).norm().item() / g0.norm().item()
    print(f"[add+LayerNorm folded:{dtype}, drop_path {drop_path}] logits rel err {le:.3e}, gradient rel-L2 {ge:.3e}")
    assert le < (1e-6 if dtype == "f32" else 1e-6)
    assert ge < (1e-5 if dtype == "f32" else 2e-2)


def test_swin_stochastic_depth_and_fused_step():
    """train mode with the reference's drop_path_rate (0.5): per-sample branches are dropped (outputs change from step to step,
    loss stays finite), the f32 step is reproducible for a fixed RNG word, the fused train_step runs and lowers the loss."""
    scfg, dcfg = SO.SWIN_TINY, SO.DEC_TINY
    model, sd = build(scfg, dcfg, "f32", 3, drop_path=0.5, dropout=0.1)
    img, expected = O.det_inputs(4, 3, scfg["img_size"], scfg["img_size"], 6, seed=40)
    imgd, expd = img.cuda(), expected.cuda()
    model.train()
    losses = []
    for _ in range(8):
        model.train_step(imgd, expd, 2e-3)
        losses.append(model.read_loss()[0])
    assert all(l == l and l < 20 for l in losses)
    assert min(losses[4:]) < losses[0]
    st = model.optimizer_state_dict()
    a = model(imgd, expd, True, 1.0).detach().clone()
    model.load_optimizer_state_dict(st)   # same RNG word -> same masks
    b = model(imgd, expd, True, 1.0).detach().clone()
    assert torch.equal(a, b)
    c = model(imgd, expd, True, 1.0).detach()
    assert not torch.equal(a, c)          # the forward after it draws new masks (the RNG word advances per training forward)
    model.eval()
    e1 = model.encode(imgd)
    e2 = model.encode(imgd)
    assert torch.equal(e1, e2)


def test_get_network_builds_swin():
    import satrn_amd
    flags = satrn_amd.Flags(dict(network="SWIN", input_size=dict(height=384, width=384),
                                 SATRN=dict(encoder=dict(hidden_dim=300, filter_dim=600, layer_num=6, head_num=8),
                                            decoder=dict(src_dim=1024, hidden_dim=512, filter_dim=512, layer_num=4, head_num=8)),
                                 data=dict(rgb=3), dropout_rate=0.1)).get()
    model = satrn_amd.get_network("SWIN", flags, None, "cuda", _DS())
    n = sum(p.numel() for p in model.parameters())
    assert 120e6 < n < 135e6   # Swin-B/384 (86.9 M) + 21841-way head (22.4 M) + decoder
    assert model.state_dict()["encoder.layers.2.blocks.17.attn_mask"].shape == (4, 144, 144)
