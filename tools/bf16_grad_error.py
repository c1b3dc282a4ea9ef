"""Where does the bf16 mode's gradient error against the f32 mode come from?  (flat gradient rel-L2 0.40, cosine 0.92 at
the benchmark batch.)  Same weights, same batch (B = 32, 1x128x384, T = 128, dropout off) through the f32 engine and the bf16 engine;
reports, per stage boundary, the error of the ACTIVATION (forward) and of the ACTIVATION GRADIENT (backward), and per parameter group
(backbone stage x tensor kind, encoder, decoder) the error of the PARAMETER gradients and each group's share of the total squared error.
Run on the GPU box:  python tools/bf16_grad_error.py [--batch 32] [--bn-eval]  -> JSON on stdout, a table on stderr."""
import argparse, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def stage_of(name):
    m = re.match(r"encoder\.shallow_cnn\.eff_block\.(\d+)\.", name)
    if m:
        return "stage" + m.group(1)
    if name.startswith("encoder.shallow_cnn.conv_stem") or name.startswith("encoder.shallow_cnn.bn1"):
        return "stem"
    if name.startswith("encoder.shallow_cnn."):
        return "conv_last"
    if name.startswith("encoder.positional_encoding"):
        return "posenc"
    m = re.match(r"encoder\.attention_layers\.(\d+)\.", name)
    if m:
        return "enc_layer" + m.group(1)
    m = re.match(r"decoder\.attention_layers\.(\d+)\.", name)
    if m:
        return "dec_layer" + m.group(1)
    return "decoder_other"


def kind_of(name, p):
    if "bn" in name.split(".")[-2] or "norm" in name.split(".")[-2]:
        return "norm." + name.split(".")[-1]
    if ".se." in name:
        return "se"
    if p.dim() == 4 and p.shape[1] == 1:
        return "depthwise"
    if name.endswith("bias"):
        return "bias"
    return "weight"


def run(dt, img, exp, H, W, bn_eval, round_w=False):
    torch.manual_seed(21)
    m = bench.make_model(dt, H, W, 0.0).to(img.device)
    if round_w:   # master weights made bf16-representable: what the bf16 mode's compute copies hold
        with torch.no_grad():
            for p_ in m.parameters():
                p_.copy_(p_.bfloat16().float())
    m.train()
    m.enable_probes(True)
    if bn_eval:
        m.eval()   # running statistics (identity at init) -- the smooth reference point: batch statistics out of the picture
        for p_ in m.parameters():
            p_.requires_grad_(True)
    logits = m(img, exp, True, 1.0) if not bn_eval else None
    if bn_eval:
        raise SystemExit("--bn-eval: use train_step(bn_eval=True) path (not wired into the probe tool)")
    acts = {k: v.clone() for k, v in m.probes().items()}
    gbuf = m.probe_grads()
    loss = m.criterion(logits.transpose(1, 2), exp[:, 1:])
    m.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    grads = {n: p_.grad.detach().float().clone() for n, p_ in m.named_parameters()}
    shapes = {n: p_ for n, p_ in m.named_parameters()}
    return dict(logits=logits.detach().float().clone(), loss=float(loss.item()), acts=acts, agrads={k: v.clone() for k, v in gbuf.items()}, grads=grads, params=shapes,
                flat=m.flat_grad().detach().float().clone())


def rel(a, b):
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def cos(a, b):
    return (torch.dot(a.flatten(), b.flatten()) / (a.norm() * b.norm()).clamp_min(1e-30)).item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--bn-eval", action="store_true")
    ap.add_argument("--split", choices=["none", "weights", "acts"], default="none",
                    help="weights: f32 engine with bf16-rounded weights against the f32 engine; acts: bf16 engine against the f32 engine, both on bf16-rounded weights")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    H, W, T, B = 128, 384, 128, a.batch
    img, exp = bench.synth(B, H, W, T, 21, dev)
    f = run("f32", img, exp, H, W, a.bn_eval, a.split == "acts")
    b = run("f32" if a.split == "weights" else "bf16", img, exp, H, W, a.bn_eval, a.split != "none")
    out = dict(config=f"EfficientSATRN B={B} 1x{H}x{W} T={T}, dropout off, training-mode BatchNorm (batch statistics); bf16 engine against the f32 engine, same weights and batch",
               loss_f32=f["loss"], loss_bf16=b["loss"], logits_rel_err=rel(b["logits"], f["logits"]),
               flat_grad_rel_l2=rel(b["flat"], f["flat"]), flat_grad_cosine=cos(b["flat"], f["flat"]))
    rows = []
    for k in f["acts"]:
        if k in b["acts"] and f["acts"][k].shape == b["acts"][k].shape:
            rows.append(dict(boundary=k, shape=list(f["acts"][k].shape), act_rel_l2=round(rel(b["acts"][k], f["acts"][k]), 5),
                             act_grad_rel_l2=round(rel(b["agrads"][k], f["agrads"][k]), 5), act_grad_cosine=round(cos(b["agrads"][k], f["agrads"][k]), 5)))
    out["stage_boundaries"] = rows
    tot_err = (b["flat"] - f["flat"]).pow(2).sum().item()
    tot_ref = f["flat"].pow(2).sum().item()
    groups = {}
    for n, gf in f["grads"].items():
        gb = b["grads"][n]
        key = (stage_of(n), kind_of(n, f["params"][n]))
        g = groups.setdefault(key, dict(err=0.0, ref=0.0, dot=0.0, nb=0.0, n=0))
        g["err"] += (gb - gf).pow(2).sum().item(); g["ref"] += gf.pow(2).sum().item(); g["dot"] += (gb * gf).sum().item(); g["nb"] += gb.pow(2).sum().item(); g["n"] += gf.numel()
    table = []
    for (st, kd), g in groups.items():
        table.append(dict(stage=st, kind=kd, numel=g["n"], rel_l2=round((g["err"] / max(g["ref"], 1e-30)) ** 0.5, 5), cosine=round(g["dot"] / max((g["ref"] * g["nb"]) ** 0.5, 1e-30), 5),
                          share_of_grad_norm_sq=round(g["ref"] / tot_ref, 5), share_of_error_sq=round(g["err"] / tot_err, 5)))
    order = ["stem"] + [f"stage{i}" for i in range(40)] + ["conv_last", "posenc", "enc_layer0", "enc_layer1", "dec_layer0", "dec_layer1", "dec_layer2", "decoder_other"]
    table.sort(key=lambda r: (order.index(r["stage"]) if r["stage"] in order else 99, r["kind"]))
    # per backbone STAGE (blocks grouped by the EfficientNetV2-S stage they belong to): 2 | 4 | 4 | 6 | 9 | 15 blocks
    out["param_groups"] = table
    print(json.dumps(out))
    err = sys.stderr
    print(f"loss f32 {f['loss']:.5f} bf16 {b['loss']:.5f}; logits rel err {out['logits_rel_err']:.3e}; flat gradient rel-L2 {out['flat_grad_rel_l2']:.3f} cosine {out['flat_grad_cosine']:.4f}", file=err)
    print(f"{'boundary':16s} {'act rel-L2':>11s} {'dAct rel-L2':>12s} {'dAct cos':>9s}", file=err)
    for r in rows:
        print(f"{r['boundary']:16s} {r['act_rel_l2']:11.4f} {r['act_grad_rel_l2']:12.4f} {r['act_grad_cosine']:9.4f}", file=err)
    print(f"{'stage':14s} {'kind':14s} {'rel-L2':>8s} {'cos':>8s} {'|g|^2 share':>12s} {'err^2 share':>12s}", file=err)
    for r in table:
        if r["share_of_error_sq"] > 0.004 or r["share_of_grad_norm_sq"] > 0.004:
            print(f"{r['stage']:14s} {r['kind']:14s} {r['rel_l2']:8.3f} {r['cosine']:8.4f} {r['share_of_grad_norm_sq']:12.4f} {r['share_of_error_sq']:12.4f}", file=err)


if __name__ == "__main__":
    main()
