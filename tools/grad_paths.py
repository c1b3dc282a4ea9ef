"""Diagnostic: the flat gradient of ONE step through the different execution paths (eager two-stream, hipGraph single chain,
segmented backward), run-to-run and path-to-path.  Usage: python tools/grad_paths.py [lite|eff] [f32|bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import satrn_oracle as O
from tests.test_model_gpu import build

net = sys.argv[1] if len(sys.argv) > 1 else "lite"
dt = sys.argv[2] if len(sys.argv) > 2 else "f32"
if net == "lite":
    cfg, H, W, B, T = dict(O.CFG_LITE), 64, 192, 4, 16
else:
    cfg, H, W, B, T = dict(O.CFG_EFF), 128, 384, 4, 32
img, exp = O.det_inputs(B, 1, H, W, T, seed=30, pad_tail=3)
img, exp = img.cuda(), exp.cuda()


def grad(mode, reps=1):
    out = []
    model, _ = build(cfg, H, W, dt, 2)
    model.train()
    for _ in range(reps):
        if mode == "eager":
            model.train_step(img, exp, 0.0, phase=1)
        elif mode == "graph":
            model.train_step(img, exp, 0.0, phase=1, use_graph=True)
        elif mode == "seg":
            model.train_step(img, exp, 0.0, phase=16 + 0 + 4 * 2)
            model.train_step(img, exp, 0.0, phase=16 + 3)
        elif mode == "seg4":
            for k in range(4):
                model.train_step(img, exp, 0.0, phase=16 + k)
        torch.cuda.synchronize()
        out.append(model.flat_grad().detach().clone())
    return out, model


def cmp(a, b, tag):
    d = (a - b).abs()
    gm = a.abs().max().item()
    rel = d / a.abs().clamp_min(1e-3 * gm)
    print(f"{tag:28s} max abs {d.max().item():.3e} (gmax {gm:.3e})  max rel(floor 1e-3 gmax) {rel.max().item():.3e}  "
          f"frac rel>1e-3: {(rel > 1e-3).float().mean().item():.5f}  frac |g|<1e-8: {(a.abs() < 1e-8).float().mean().item():.4f}")
    return rel


def per_tensor(a, b, model, tag, k=6):
    rows = []
    for e in model._entries:
        if e[1] != 0:
            continue
        x, y = a[e[3]:e[3] + e[4]], b[e[3]:e[3] + e[4]]
        rows.append(((x - y).abs().max().item() / max(x.abs().max().item(), 1e-30), x.abs().max().item(), e[0]))
    rows.sort(reverse=True)
    print(tag, "worst tensors (max|diff| / max|g|, max|g|):")
    for r in rows[:k]:
        print(f"    {r[0]:.3e}  {r[1]:.3e}  {r[2]}")


ge, me = grad("eager", 3)
cmp(ge[0], ge[1], "eager run0 vs run1")
per_tensor(ge[0], ge[1], me, "run0 vs run1")
per_tensor(ge[1], ge[2], me, "run1 vs run2")
cmp(ge[0], ge[2], "eager run0 vs run2")
gg, _ = grad("graph", 3)
cmp(ge[0], gg[0], "eager vs graph(first=eager)")
cmp(ge[0], gg[1], "eager vs graph replay1")
cmp(gg[1], gg[2], "graph replay1 vs replay2")
gs, _ = grad("seg", 2)
cmp(ge[0], gs[0], "eager vs seg(0-2|3)")
cmp(gs[0], gs[1], "seg run0 vs run1")
g4, _ = grad("seg4", 2)
rel = cmp(ge[0], g4[0], "eager vs seg4")
# where do the large differences sit?
worst = torch.nonzero(rel > 1e-3).flatten()
if worst.numel():
    ents = [(e[0], e[3], e[4]) for e in me._entries if e[1] == 0]
    import collections
    c = collections.Counter()
    for i in worst[:20000].tolist():
        for name, off, n in ents:
            if off <= i < off + n:
                c[name] += 1
                break
    print(c.most_common(12))
