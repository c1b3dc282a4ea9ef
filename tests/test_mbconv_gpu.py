"""GPU: the MBConv block kernels (kernels_mbconv.hip) through the C-ABI -- the front of a timm MBConv block (expand product, BatchNorm +
SiLU, depthwise 3x3, BatchNorm + SiLU, squeeze-and-excite; SURVEY Appendix B stages 3-5, networks/EfficientSATRN.py:74,84) as ONE launch --
against an fp32 torch restatement with the same bf16 rounding points AND against the three operators it replaces, at the shapes of
the benchmark configuration (32 images, 4x12 x 1536 and 8x24 x 960 channels) and at small / ragged batch sizes."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from tests.test_ops_gpu import P, close, dev, lib, nchw, nhwc, ok, pack_dense, q, rnd, st, _keepalive  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu
BF = "bf16"


def _mailbox(B, C):
    return torch.zeros(3 * (C // 64) * B * 128 + B * (C // 64) * 64, dtype=torch.int64, device="cuda")


def _inputs(B, H, W, Cin, C, S):
    x = q(rnd(B, H * W, Cin, seed=1), BF)
    W0 = q(rnd(C, Cin, seed=2, scale=Cin ** -0.5 * 1.7), BF)
    g1, be1 = 1 + rnd(C, seed=3, scale=0.2), rnd(C, seed=4, scale=0.1)
    g2, be2 = 1 + rnd(C, seed=5, scale=0.2), rnd(C, seed=6, scale=0.1)
    dw = q(rnd(C, 1, 3, 3, seed=7, scale=0.4), BF)
    W1, W2 = q(rnd(S, C, seed=8, scale=0.05), BF), q(rnd(C, S, seed=9, scale=0.2), BF)
    b1, b2 = rnd(S, seed=10, scale=0.1), rnd(C, seed=11, scale=0.1)
    return x, W0, g1, be1, g2, be2, dw, W1, W2, b1, b2


def _reference(x, W0, g1, be1, g2, be2, dw, W1, W2, b1, b2, B, H, W, C, eps):
    """fp32 with the engine's rounding points: y1, z1, y2, z2, gate and z3 are stored in bf16; the statistics come from the f32 products"""
    M = B * H * W
    y1f = x.reshape(M, -1) @ W0.t()
    m1, v1 = y1f.mean(0), y1f.var(0, unbiased=False)
    y1 = q(y1f, BF)
    sc1 = g1 / torch.sqrt(v1 + eps)
    z1 = q(F.silu(y1 * sc1 + (be1 - m1 * sc1)), BF)
    y2f = nhwc(F.conv2d(nchw(z1.reshape(B, H, W, C)), dw, None, 1, 1, 1, C)).reshape(M, C)
    m2, v2 = y2f.mean(0), y2f.var(0, unbiased=False)
    y2 = q(y2f, BF)
    sc2 = g2 / torch.sqrt(v2 + eps)
    z2 = q(F.silu(y2 * sc2 + (be2 - m2 * sc2)), BF)
    pooled = z2.reshape(B, H * W, C).mean(1)
    u1 = pooled @ W1.t() + b1
    s1 = F.silu(u1)
    gate = q(torch.sigmoid(s1 @ W2.t() + b2), BF)
    z3 = q(z2.reshape(B, H * W, C) * gate[:, None, :], BF)
    return dict(y1=y1, z1=z1, y2=y2, z2=z2, pooled=pooled, u1=u1, s1=s1, gate=gate, z3=z3.reshape(M, C), m1=m1, v1=v1, m2=m2, v2=v2, sc1=sc1, sc2=sc2)


def _run_front(lib, inp, B, H, W, Cin, C, S, eps, box, keep_z2=1):
    x, W0, g1, be1, g2, be2, dw, W1, W2, b1, b2 = inp
    M = B * H * W
    bf = torch.bfloat16
    W0d, _, _ = pack_dense(lib, W0, BF)
    wp = torch.empty(9, C, dtype=bf, device="cuda")
    ok(lib, lib.satrn_pack_dwconv3x3(1, P(dev(dw)), P(wp), C, st()))
    o = dict(y1=torch.zeros(M, C, dtype=bf, device="cuda"), z1=torch.zeros(M, C, dtype=bf, device="cuda"), y2=torch.zeros(M, C, dtype=bf, device="cuda"),
             z2=torch.zeros(M, C, dtype=bf, device="cuda"), z3=torch.zeros(M, C, dtype=bf, device="cuda"), pooled=torch.zeros(B, C, device="cuda"),
             u1=torch.zeros(B, S, device="cuda"), s1=torch.zeros(B, S, device="cuda"), gate=torch.zeros(B, C, dtype=bf, device="cuda"),
             coef1=torch.zeros(4 * C, device="cuda"), coef2=torch.zeros(4 * C, device="cuda"),
             rm1=torch.zeros(C, device="cuda"), rv1=torch.ones(C, device="cuda"), rm2=torch.zeros(C, device="cuda"), rv2=torch.ones(C, device="cuda"),
             nbt1=torch.zeros(1, dtype=torch.int64, device="cuda"), nbt2=torch.zeros(1, dtype=torch.int64, device="cuda"))
    rc = lib.satrn_mbconv_front_fwd(P(dev(x, BF)), P(W0d), P(o["y1"]), P(dev(g1)), P(dev(be1)), P(o["rm1"]), P(o["rv1"]), P(o["nbt1"]), P(o["coef1"]),
                                    P(o["z1"]), P(wp), P(o["y2"]), P(dev(g2)), P(dev(be2)), P(o["rm2"]), P(o["rv2"]), P(o["nbt2"]), P(o["coef2"]),
                                    P(o["z2"]), keep_z2, P(dev(W1, BF)), P(dev(b1)), P(dev(W2, BF)), P(dev(b2)), P(o["pooled"]), P(o["u1"]), P(o["s1"]),
                                    P(o["gate"]), P(o["z3"]), B, H, W, Cin, C, S, eps, P(box), box.numel(), st())
    torch.cuda.synchronize()
    return rc, o


@pytest.mark.parametrize("B,H,W,Cin,C,S", [(32, 4, 12, 256, 1536, 64), (32, 8, 24, 160, 960, 40), (5, 8, 24, 128, 512, 32), (3, 4, 12, 256, 1536, 64),
                                           (1, 4, 12, 256, 128, 8), (17, 8, 24, 160, 192, 16)])
def test_mbconv_front_one_launch_vs_torch(lib, B, H, W, Cin, C, S):
    eps = 1e-3
    inp = _inputs(B, H, W, Cin, C, S)
    ref = _reference(*inp, B, H, W, C, eps)
    box = _mailbox(B, C)
    M = B * H * W
    for rep in range(3):   # the mailbox is reused as it is (a launch number tags every word)
        rc, o = _run_front(lib, inp, B, H, W, Cin, C, S, eps, box, keep_z2=rep % 2)
        assert rc == 0, lib.satrn_last_error().decode()
        assert lib.satrn_device_error(st()) == 0
        close(o["y1"], ref["y1"], BF, "front y1", bf16_tol=1e-2)
        close(o["coef1"][2 * C:3 * C], ref["m1"], "f32", "front mean 1", f32_tol=2e-4)
        close(o["coef1"][:C], ref["sc1"], "f32", "front scale 1", f32_tol=1e-3)
        close(o["z1"], ref["z1"], BF, "front z1", bf16_tol=2e-2)
        close(o["y2"], ref["y2"], BF, "front y2", bf16_tol=2e-2)
        close(o["coef2"][2 * C:3 * C], ref["m2"], "f32", "front mean 2", f32_tol=5e-3)
        close(o["coef2"][:C], ref["sc2"], "f32", "front scale 2", f32_tol=1e-2)
        close(o["pooled"], ref["pooled"], BF, "front pooled", bf16_tol=2e-2)
        close(o["u1"], ref["u1"], BF, "front u1", bf16_tol=2e-2)
        close(o["gate"], ref["gate"], BF, "front gate", bf16_tol=1e-2)
        close(o["z3"], ref["z3"], BF, "front z3", bf16_tol=3e-2)
        if rep % 2:
            close(o["z2"], ref["z2"], BF, "front z2", bf16_tol=3e-2)
        assert o["nbt1"].item() == 1 and o["nbt2"].item() == 1
        unb = M / (M - 1)
        close(o["rm1"], 0.1 * ref["m1"], "f32", "front running mean 1", f32_tol=1e-3)
        close(o["rv1"], 0.9 + 0.1 * ref["v1"] * unb, "f32", "front running var 1", f32_tol=1e-3)
        close(o["rv2"], 0.9 + 0.1 * ref["v2"] * unb, "f32", "front running var 2", f32_tol=1e-2)


@pytest.mark.parametrize("B,H,W,Cin,C,S", [(32, 4, 12, 256, 1536, 64), (32, 8, 24, 160, 960, 40), (7, 8, 24, 128, 512, 32)])
def test_mbconv_front_one_launch_vs_the_three_operators(lib, B, H, W, Cin, C, S):
    """the one launch against satrn_linear_fwd_stats -> satrn_batchnorm_act_dwconv3x3_fwd -> satrn_batchnorm_act_se_fwd on the same inputs:
    same rounding points, so the two forms differ only through the summation order of the statistics (and of the k loop)."""
    eps = 1e-3
    inp = _inputs(B, H, W, Cin, C, S)
    x, W0, g1, be1, g2, be2, dw, W1, W2, b1, b2 = inp
    M = B * H * W
    box = _mailbox(B, C)
    rc, o = _run_front(lib, inp, B, H, W, Cin, C, S, eps, box)
    assert rc == 0, lib.satrn_last_error().decode()
    bf = torch.bfloat16
    W0d, _, _ = pack_dense(lib, W0, BF)
    wp = torch.empty(9, C, dtype=bf, device="cuda")
    ok(lib, lib.satrn_pack_dwconv3x3(1, P(dev(dw)), P(wp), C, st()))
    y1 = torch.zeros(M, C, dtype=bf, device="cuda")
    stats1 = torch.zeros(2 * C, device="cuda")
    ok(lib, lib.satrn_linear_fwd_stats(1, P(dev(x, BF)), P(W0d), P(y1), M, C, Cin, P(stats1), 1, None, None, None, 0, 0, st()))
    scr1b, stats2 = torch.zeros(6 * C, device="cuda"), torch.zeros(2 * C, device="cuda")
    z1, y2 = torch.zeros(M, C, dtype=bf, device="cuda"), torch.zeros(M, C, dtype=bf, device="cuda")
    rm, rv, nbt = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), torch.zeros(1, dtype=torch.int64, device="cuda")
    # (the operator recomputes the column sums of y1 itself: they equal the epilogue's up to summation order)
    ok(lib, lib.satrn_batchnorm_act_dwconv3x3_fwd(1, P(y1), P(dev(g1)), P(dev(be1)), P(rm), P(rv), P(nbt), eps, 2, P(z1), P(wp), None, P(y2), P(stats2),
                                                  B, H, W, C, P(scr1b), st()))
    scr2 = torch.zeros(6 * C, device="cuda")
    z2, z3 = torch.zeros(M, C, dtype=bf, device="cuda"), torch.zeros(M, C, dtype=bf, device="cuda")
    po, u1, s1 = torch.zeros(B, C, device="cuda"), torch.zeros(B, S, device="cuda"), torch.zeros(B, S, device="cuda")
    gd = torch.zeros(B, C, dtype=bf, device="cuda")
    box2 = torch.zeros(128 * 1600, dtype=torch.int64, device="cuda")
    rm2, rv2, nbt2 = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), torch.zeros(1, dtype=torch.int64, device="cuda")
    ok(lib, lib.satrn_batchnorm_act_se_fwd(1, P(y2), P(dev(g2)), P(dev(be2)), P(rm2), P(rv2), P(nbt2), eps, 2, P(z2), 1, P(dev(W1, BF)),
                                           P(dev(b1)), P(dev(W2, BF)), P(dev(b2)), P(po), P(u1), P(s1), P(gd), P(z3), B, H * W, C, S, P(scr2), P(box2), 128, st()))
    torch.cuda.synchronize()
    for name, a, c in (("y1", o["y1"], y1), ("z1", o["z1"], z1), ("y2", o["y2"], y2), ("z3", o["z3"], z3)):
        same = (a == c).float().mean().item()
        print(f"[front vs three operators {name}] identical elements {same:.6f}")
        close(a, c.cpu(), BF, f"front vs three operators {name}", bf16_tol=1.6e-2)
        assert same > (0.999 if name == "y1" else 0.9)   # (statistics summed in another order: a last-bit change of scale / shift flips some roundings)
    close(o["gate"], gd.cpu(), BF, "front vs three operators gate", bf16_tol=1e-2)
    close(o["pooled"], po.cpu(), BF, "front vs three operators pooled", bf16_tol=1e-2)


def test_mbconv_front_refuses_what_it_cannot_hold(lib):
    """shapes outside the one-launch form return -1 and touch nothing: a map that is not 48 / 192 pixels, a batch beyond the mailbox"""
    box = _mailbox(4, 128)
    inp = _inputs(4, 6, 10, 128, 128, 8)
    rc, _ = _run_front(lib, inp, 4, 6, 10, 128, 128, 8, 1e-3, box)
    assert rc == -1
    inp = _inputs(65, 4, 12, 256, 64, 8)
    rc, _ = _run_front(lib, inp, 65, 4, 12, 256, 64, 8, 1e-3, _mailbox(65, 64))
    assert rc == -1
    assert lib.satrn_device_error(st()) == 0


@pytest.mark.parametrize("B,H,W,Cout,C,S", [(32, 4, 12, 256, 1536, 64), (32, 8, 24, 160, 960, 40), (6, 8, 24, 128, 512, 32), (3, 4, 12, 256, 192, 16), (1, 4, 12, 256, 64, 8)])
def test_mbconv_backward_projection_data_gradient_and_squeeze_excite_in_one_launch(lib, B, H, W, Cout, C, S):
    """satrn_mbconv_bwd_se against (a) the two operators it replaces (satrn_linear_bwd_data, then satrn_se_bwd_bnred on the data gradient
    that one stored) and (b) torch on the formulas; the forward state (BatchNorm 2 coefficients, gate, u1) comes from the device forward."""
    bf, HW, M, eps = torch.bfloat16, H * W, B * H * W, 1e-3
    y2 = q(rnd(M, C, seed=1) * 2 + 0.5, BF)
    w, b = 1 + rnd(C, seed=3, scale=0.2), rnd(C, seed=4, scale=0.1)
    W1, W2 = q(rnd(S, C, seed=12, scale=0.05), BF), q(rnd(C, S, seed=13, scale=0.2), BF)
    b1, b2 = rnd(S, seed=14, scale=0.1), rnd(C, seed=15, scale=0.1)
    Wp = q(rnd(Cout, C, seed=16, scale=C ** -0.5 * 2), BF)     # the projection [Cout][C]
    dy3 = q(rnd(M, Cout, seed=6), BF)
    y2d = dev(y2, BF)
    rmd, rvd, nbt = dev(torch.zeros(C)), dev(torch.ones(C)), torch.zeros(1, dtype=torch.int64, device="cuda")
    scratch = torch.zeros(6 * C, device="cuda")
    z2d = torch.empty(M, C, dtype=bf, device="cuda")
    ok(lib, lib.satrn_batchnorm_act_fwd(1, P(y2d), P(dev(w)), P(dev(b)), P(rmd), P(rvd), P(nbt), eps, 1, 2, None, P(z2d), M, C, P(scratch), st()))
    W1d, W2d = dev(W1, BF), dev(W2, BF)
    pooled_d, u1_d, s1_d = (torch.zeros(B, n, device="cuda") for n in (C, S, S))
    gate_d = torch.zeros(B, C, dtype=bf, device="cuda")
    z3d = torch.zeros(B, HW, C, dtype=bf, device="cuda")
    ok(lib, lib.satrn_se_fwd(1, P(z2d), P(W1d), P(dev(b1)), P(W2d), P(dev(b2)), None, P(pooled_d), P(u1_d), P(s1_d), P(gate_d), P(z3d), B, HW, C, S, st()))
    _, Wbd, ldb = pack_dense(lib, Wp, BF)
    dy3d = dev(dy3, BF)
    # (a) the two operators
    dz3_a = torch.zeros(M, C, dtype=bf, device="cuda")
    ok(lib, lib.satrn_linear_bwd_data(1, P(dy3d), Cout, P(Wbd), ldb, P(dz3_a), M, Cout, C, 0, st()))
    dz2_a, du1_a, ds1_a = torch.zeros(B, C, device="cuda"), torch.zeros(B, S, device="cuda"), torch.zeros(B, S, device="cuda")
    dpool_a = torch.zeros(B, C, dtype=bf, device="cuda")
    Pd, red_a = torch.zeros(4 * B * C, device="cuda"), torch.zeros(2 * C, device="cuda")
    ok(lib, lib.satrn_se_bwd_bnred(1, P(dz3_a), P(y2d), P(scratch), 2, P(gate_d), P(u1_d), P(W1d), P(W2d), P(dz2_a), P(du1_a), P(ds1_a), P(dpool_a),
                                   P(Pd), P(red_a), B, HW, C, S, st()))
    # the one launch, three times on a never-cleared mailbox
    box = torch.zeros(B * (C // 64) * 64, dtype=torch.int64, device="cuda")
    prev = None
    for rep in range(3):
        dz3 = torch.zeros(M, C, dtype=bf, device="cuda")
        dz2, du1, ds1 = torch.zeros(B, C, device="cuda"), torch.zeros(B, S, device="cuda"), torch.full((B, S), 7.0, device="cuda")
        dpool, red = torch.zeros(B, C, dtype=bf, device="cuda"), torch.zeros(2 * C, device="cuda")
        rc = lib.satrn_mbconv_bwd_se(P(dy3d), P(Wbd), ldb, P(dz3), P(y2d), P(scratch[2 * C:]), P(gate_d), P(u1_d), P(W1d), P(W2d), P(dz2), P(ds1), P(du1), P(dpool),
                                     P(red), B, H, W, Cout, C, S, P(box), box.numel(), st())
        assert rc == 0, lib.satrn_last_error().decode()
        torch.cuda.synchronize()
        assert lib.satrn_device_error(st()) == 0
        same = (dz3 == dz3_a).float().mean().item()
        print(f"[bwd se one launch] dz3 identical to the product's {same:.6f}")
        close(dz3, dz3_a.float().cpu(), BF, "dz3 vs linear_bwd_data", bf16_tol=1e-2)
        close(dz2, dz2_a.cpu(), BF, "dz2 vs se_bwd_bnred", bf16_tol=1e-2)
        close(ds1, ds1_a.cpu(), BF, "ds1 vs se_bwd_bnred", bf16_tol=1e-2)
        close(du1, du1_a.cpu(), BF, "du1 vs se_bwd_bnred", bf16_tol=1e-2)
        close(dpool, dpool_a.float().cpu(), BF, "dpooled vs se_bwd_bnred", bf16_tol=1.5e-2)
        close(red, red_a.cpu(), BF, "BatchNorm 2 sums vs se_bwd_bnred", bf16_tol=1e-2)
        if prev is not None:
            for u_, v_, what in zip((dz3, dz2, ds1, du1, dpool), prev, ("dz3", "dz2", "ds1", "du1", "dpooled")):
                assert torch.equal(u_, v_), f"{what}: differs from call to call"
        prev = (dz3, dz2, ds1, du1, dpool)
    # (b) torch on the formulas, from the device's forward state
    dz3_r = q(dy3 @ Wp, BF).reshape(B, HW, C)
    z2 = z2d.float().cpu().reshape(B, HW, C)
    gate, u1 = gate_d.float().cpu(), u1_d.cpu()
    dgate = (dz3_r * z2).sum(1)
    dz2_r = dgate * gate * (1 - gate)
    ds1_r = dz2_r @ W2
    sg = torch.sigmoid(u1)
    du1_r = ds1_r * (sg * (1 + u1 * (1 - sg)))
    dpool_r = du1_r @ W1
    close(dz3, dz3_r.reshape(M, C), BF, "dz3 vs torch", bf16_tol=1e-2)
    close(dz2, dz2_r, BF, "dz2 vs torch", bf16_tol=2e-2)
    close(du1, du1_r, BF, "du1 vs torch", bf16_tol=2e-2)
    close(dpool, dpool_r, BF, "dpooled vs torch", bf16_tol=2e-2)


@pytest.mark.parametrize("B,H,W,Cin,C,S,res", [(32, 4, 12, 256, 1536, 64, True), (32, 8, 24, 160, 960, 40, True), (5, 8, 24, 128, 512, 32, False), (3, 4, 12, 256, 128, 8, True)])
def test_mbconv_front_with_the_input_batchnorm_folded_in(lib, B, H, W, Cin, C, S, res):
    """satrn_mbconv_front_fwd_bn_in: the block input = BatchNorm(in_y) (+ residual), normalised while the kernel stages it -- against
    satrn_batchnorm_act_fwd followed by satrn_mbconv_front_fwd on the tensor that one wrote (same coefficients, same rounding point: the
    block then runs on bit-identical inputs)."""
    eps, M, bf = 1e-3, B * H * W, torch.bfloat16
    inp = _inputs(B, H, W, Cin, C, S)
    _, W0, g1, be1, g2, be2, dw, W1, W2, b1, b2 = inp
    in_y = q(rnd(M, Cin, seed=21) * 1.5 + 0.3, BF)
    in_res = q(rnd(M, Cin, seed=22), BF) if res else None
    gw, gb = 1 + rnd(Cin, seed=23, scale=0.2), rnd(Cin, seed=24, scale=0.1)
    rep = 3
    sums = torch.zeros(rep, 2, Cin)
    yf = in_y.float()
    for r in range(rep):   # the column sums split over replicas, as a statistics epilogue leaves them
        rows = yf[r::rep]
        sums[r, 0], sums[r, 1] = rows.sum(0), (rows * rows).sum(0)
    in_yd, in_resd, sumsd = dev(in_y, BF), (dev(in_res, BF) if res else None), dev(sums)
    # reference route: the BatchNorm as its own launch, then the block on its output
    rm, rv, nbt = torch.zeros(Cin, device="cuda"), torch.ones(Cin, device="cuda"), torch.zeros(1, dtype=torch.int64, device="cuda")
    scr = torch.zeros(6 * Cin, device="cuda")   # (the operator sums the columns of in_y itself)
    xd = torch.zeros(M, Cin, dtype=bf, device="cuda")
    ok(lib, lib.satrn_batchnorm_act_fwd(1, P(in_yd), P(dev(gw)), P(dev(gb)), P(rm), P(rv), P(nbt), eps, 1, 0, P(in_resd), P(xd), M, Cin, P(scr), st()))
    torch.cuda.synchronize()
    inp2 = (xd.float().cpu().reshape(B, H * W, Cin),) + inp[1:]
    box = _mailbox(B, C)
    rc, o_ref = _run_front(lib, inp2, B, H, W, Cin, C, S, eps, box)
    assert rc == 0, lib.satrn_last_error().decode()
    # folded route
    W0d, _, _ = pack_dense(lib, W0, BF)
    wp = torch.empty(9, C, dtype=bf, device="cuda")
    ok(lib, lib.satrn_pack_dwconv3x3(1, P(dev(dw)), P(wp), C, st()))
    o = dict(y1=torch.zeros(M, C, dtype=bf, device="cuda"), z1=torch.zeros(M, C, dtype=bf, device="cuda"), y2=torch.zeros(M, C, dtype=bf, device="cuda"),
             z2=torch.zeros(M, C, dtype=bf, device="cuda"), z3=torch.zeros(M, C, dtype=bf, device="cuda"), pooled=torch.zeros(B, C, device="cuda"),
             u1=torch.zeros(B, S, device="cuda"), s1=torch.zeros(B, S, device="cuda"), gate=torch.zeros(B, C, dtype=bf, device="cuda"),
             coef1=torch.zeros(4 * C, device="cuda"), coef2=torch.zeros(4 * C, device="cuda"))
    rm1, rv1, rm2, rv2 = (torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"))
    n1, n2, n0 = (torch.zeros(1, dtype=torch.int64, device="cuda") for _ in range(3))
    irm, irv, icoef = torch.zeros(Cin, device="cuda"), torch.ones(Cin, device="cuda"), torch.zeros(4 * Cin, device="cuda")
    xo = torch.zeros(M, Cin, dtype=bf, device="cuda")
    rc = lib.satrn_mbconv_front_fwd_bn_in(P(in_yd), P(in_resd), P(sumsd), rep, P(dev(gw)), P(dev(gb)), P(irm), P(irv), P(n0), P(icoef), P(xo), P(W0d), P(o["y1"]),
                                          P(dev(g1)), P(dev(be1)), P(rm1), P(rv1), P(n1), P(o["coef1"]), P(o["z1"]), P(wp), P(o["y2"]), P(dev(g2)), P(dev(be2)),
                                          P(rm2), P(rv2), P(n2), P(o["coef2"]), P(o["z2"]), 1, P(dev(W1, BF)), P(dev(b1)), P(dev(W2, BF)), P(dev(b2)),
                                          P(o["pooled"]), P(o["u1"]), P(o["s1"]), P(o["gate"]), P(o["z3"]), B, H, W, Cin, C, S, eps, P(box), box.numel(), st())
    assert rc == 0, lib.satrn_last_error().decode()
    torch.cuda.synchronize()
    assert lib.satrn_device_error(st()) == 0
    same_x = (xo == xd).float().mean().item()
    print(f"[front with folded input BatchNorm] x identical to batchnorm_act_fwd's {same_x:.6f}")
    close(xo, xd.float().cpu(), BF, "folded input x", bf16_tol=1e-2)
    assert same_x > 0.99   # (the replicas are added in another order: a last-bit difference of scale / shift may move a rounding)
    close(icoef, scr[2 * Cin:].cpu(), "f32", "input BatchNorm coefficients", f32_tol=1e-5)
    close(irm, rm.cpu(), "f32", "input BatchNorm running mean", f32_tol=1e-5)
    close(irv, rv.cpu(), "f32", "input BatchNorm running var", f32_tol=1e-5)
    assert n0.item() == 1
    for k in ("y1", "z1", "y2", "z3"):
        close(o[k], o_ref[k].float().cpu(), BF, f"folded input: {k}", bf16_tol=2e-2)
    close(o["gate"], o_ref["gate"].float().cpu(), BF, "folded input: gate", bf16_tol=1e-2)


@pytest.mark.parametrize("B,H,W,Cout,C,S", [(32, 4, 12, 256, 1536, 64), (32, 8, 24, 160, 960, 40), (6, 8, 24, 128, 512, 32), (2, 4, 12, 256, 128, 8)])
def test_mbconv_backward_with_the_closing_batchnorm_backward_folded_in(lib, B, H, W, Cout, C, S):
    """satrn_mbconv_bwd_se_bn_in: dy3 = backward of the block-ending BatchNorm applied while the kernel stages it -- against
    satrn_batchnorm_act_bwd_apply followed by satrn_mbconv_bwd_se on the dy3 that one wrote."""
    bf, HW, M, eps = torch.bfloat16, H * W, B * H * W, 1e-3
    y2 = q(rnd(M, C, seed=1) * 2 + 0.5, BF)
    w, b = 1 + rnd(C, seed=3, scale=0.2), rnd(C, seed=4, scale=0.1)
    W1, W2 = q(rnd(S, C, seed=12, scale=0.05), BF), q(rnd(C, S, seed=13, scale=0.2), BF)
    b1, b2 = rnd(S, seed=14, scale=0.1), rnd(C, seed=15, scale=0.1)
    Wp = q(rnd(Cout, C, seed=16, scale=C ** -0.5 * 2), BF)
    y3 = q(rnd(M, Cout, seed=17) * 1.3 + 0.2, BF)
    w3, b3 = 1 + rnd(Cout, seed=18, scale=0.2), rnd(Cout, seed=19, scale=0.1)
    dz = q(rnd(M, Cout, seed=6), BF)
    y2d, y3d, dzd = dev(y2, BF), dev(y3, BF), dev(dz, BF)
    # forward state of bn2 / se (as in the test above) and of bn3
    rmd, rvd, nbt = dev(torch.zeros(C)), dev(torch.ones(C)), torch.zeros(1, dtype=torch.int64, device="cuda")
    scr2 = torch.zeros(6 * C, device="cuda")
    z2d = torch.empty(M, C, dtype=bf, device="cuda")
    ok(lib, lib.satrn_batchnorm_act_fwd(1, P(y2d), P(dev(w)), P(dev(b)), P(rmd), P(rvd), P(nbt), eps, 1, 2, None, P(z2d), M, C, P(scr2), st()))
    W1d, W2d = dev(W1, BF), dev(W2, BF)
    pooled_d, u1_d, s1_d = (torch.zeros(B, n, device="cuda") for n in (C, S, S))
    gate_d = torch.zeros(B, C, dtype=bf, device="cuda")
    z3d = torch.zeros(B, HW, C, dtype=bf, device="cuda")
    ok(lib, lib.satrn_se_fwd(1, P(z2d), P(W1d), P(dev(b1)), P(W2d), P(dev(b2)), None, P(pooled_d), P(u1_d), P(s1_d), P(gate_d), P(z3d), B, HW, C, S, st()))
    rm3, rv3 = dev(torch.zeros(Cout)), dev(torch.ones(Cout))
    scr3 = torch.zeros(6 * Cout, device="cuda")
    outd = torch.empty(M, Cout, dtype=bf, device="cuda")
    ok(lib, lib.satrn_batchnorm_act_fwd(1, P(y3d), P(dev(w3)), P(dev(b3)), P(rm3), P(rv3), P(nbt), eps, 1, 0, None, P(outd), M, Cout, P(scr3), st()))
    # bn3's backward sums from torch (sum dz, sum dz * xhat)
    mu, rs = scr3[4 * Cout:5 * Cout].cpu(), scr3[5 * Cout:6 * Cout].cpu()
    xh = (y3 - mu) * rs
    sums3 = dev(torch.cat([dz.sum(0), (dz * xh).sum(0)]))
    _, Wbd, ldb = pack_dense(lib, Wp, BF)
    # reference route: apply pass, then the one-launch backward on its output
    dy3_a = torch.zeros(M, Cout, dtype=bf, device="cuda")
    dw_a, db_a = torch.zeros(Cout, device="cuda"), torch.zeros(Cout, device="cuda")
    ok(lib, lib.satrn_batchnorm_act_bwd_apply(1, P(dzd), P(y3d), P(dev(w3)), P(scr3), 0, P(dy3_a), P(dw_a), P(db_a), M, Cout, P(sums3), st()))
    box = torch.zeros(B * (C // 64) * 64, dtype=torch.int64, device="cuda")

    def outs():
        return dict(dz3=torch.zeros(M, C, dtype=bf, device="cuda"), dz2=torch.zeros(B, C, device="cuda"), ds1=torch.zeros(B, S, device="cuda"),
                    du1=torch.zeros(B, S, device="cuda"), dpool=torch.zeros(B, C, dtype=bf, device="cuda"), red=torch.zeros(2 * C, device="cuda"))
    a = outs()
    rc = lib.satrn_mbconv_bwd_se(P(dy3_a), P(Wbd), ldb, P(a["dz3"]), P(y2d), P(scr2[2 * C:]), P(gate_d), P(u1_d), P(W1d), P(W2d), P(a["dz2"]), P(a["ds1"]), P(a["du1"]),
                                 P(a["dpool"]), P(a["red"]), B, H, W, Cout, C, S, P(box), box.numel(), st())
    assert rc == 0, lib.satrn_last_error().decode()
    f = outs()
    dy3_f = torch.zeros(M, Cout, dtype=bf, device="cuda")
    dw_f, db_f = torch.zeros(Cout, device="cuda"), torch.zeros(Cout, device="cuda")
    rc = lib.satrn_mbconv_bwd_se_bn_in(P(dzd), P(y3d), P(scr3[2 * Cout:]), P(dev(w3)), P(sums3), P(dy3_f), P(dw_f), P(db_f), P(Wbd), ldb, P(f["dz3"]), P(y2d),
                                       P(scr2[2 * C:]), P(gate_d), P(u1_d), P(W1d), P(W2d), P(f["dz2"]), P(f["ds1"]), P(f["du1"]), P(f["dpool"]), P(f["red"]),
                                       B, H, W, Cout, C, S, P(box), box.numel(), st())
    assert rc == 0, lib.satrn_last_error().decode()
    torch.cuda.synchronize()
    assert lib.satrn_device_error(st()) == 0
    same = (dy3_f == dy3_a).float().mean().item()
    print(f"[bwd with folded BatchNorm backward] dy3 identical to batchnorm_act_bwd_apply's {same:.6f}")
    assert same > 0.999
    close(dy3_f, dy3_a.float().cpu(), BF, "folded dy3", bf16_tol=1e-2)
    close(dw_f, dw_a.cpu(), "f32", "bn3 dweight", f32_tol=1e-5)
    close(db_f, db_a.cpu(), "f32", "bn3 dbias", f32_tol=1e-5)
    for k in ("dz3", "dz2", "ds1", "du1", "dpool", "red"):
        close(f[k], a[k].float().cpu(), BF, f"folded: {k}", bf16_tol=1e-2)
