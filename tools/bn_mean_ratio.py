"""|batch mean| / batch std per channel of every BatchNorm INPUT after one training step from the benchmark's initial weights, read off the running
statistics (momentum 0.1 from mean 0 / var 1): how much of a pre-BatchNorm tensor's bf16 rounding error is amplified by its mean offset
(rounding error ~ 2^-9 |y|, what matters to the normalised value is error / sigma).  GPU box: python tools/bn_mean_ratio.py"""
import os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
torch.manual_seed(21)
m = bench.make_model("bf16", 128, 384, 0.0).to(dev)
m.train()
img, exp = bench.synth(32, 128, 384, 128, 21, dev)
m.train_step(img, exp, 0.0)
torch.cuda.synchronize()
sd = m.state_dict()
rows = {}
for k, v in sd.items():
    if k.endswith("running_mean"):
        rv = sd[k[:-len("running_mean")] + "running_var"].float().cpu()
        mean = v.float().cpu() * 10.0
        var = ((rv - 0.9) * 10.0).clamp_min(1e-12)
        r = mean.abs() / var.sqrt()
        mm = re.match(r"encoder\.shallow_cnn\.eff_block\.(\d+)\.(\d+)\.(\w+)", k)
        grp = f"stage{mm.group(1)} {mm.group(3)}" if mm else k.split(".running")[0][-40:]
        rows.setdefault(grp, []).append(r)
print(f"{'BatchNorm input':44s} {'median |mean|/std':>18s} {'p90':>8s} {'max':>8s}")
for g, rs in rows.items():
    r = torch.cat(rs)
    print(f"{g:44s} {r.median().item():18.3f} {r.quantile(0.9).item():8.3f} {r.max().item():8.3f}")
