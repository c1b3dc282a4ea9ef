"""Generate golden fixtures by running THE REFERENCE ITSELF on CPU (authoring container only).

Usage (from the repo root, in the container that has /root/reference):
    python tests/golden/make_golden.py

Imports the reference's own ``networks`` package from /root/reference with the harness of
SURVEY.md Appendix C (stub modules for absent third-party packages, ``utils`` imported first,
``Tensor.get_device`` patched for CPU), loads build-owned deterministic weights
(``oracle.satrn_oracle.det_state_dict``) into the reference modules and stores inputs' seeds and
the reference's outputs in ``tests/golden/*.npz``.  The reference never travels: only these
vectors (data) are committed.

timm is not installable here, so ``timm.create_model`` is shimmed to return an object whose
``.blocks`` is the build's own restatement of the EfficientNetV2-S blocks (parity of the blocks
themselves is UNPINNED; everything around them is the reference's code).
"""
import os
import sys
import types
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
from oracle import satrn_oracle as O  # noqa: E402

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


# ---------------------------------------------------------------- reference import harness
class _ShimBlock(nn.Module):
    def __init__(self, desc, specs, prefix):
        super().__init__()
        self.desc = desc
        self._names = []
        for name, (shape, kind) in specs.items():
            if not name.startswith(prefix):
                continue
            rel = name[len(prefix):]
            mod = self
            parts = rel.split(".")
            for p in parts[:-1]:
                if not hasattr(mod, p):
                    mod.add_module(p, nn.Module())
                mod = getattr(mod, p)
            if kind in ("bn_rm", "bn_rv"):
                mod.register_buffer(parts[-1], torch.zeros(shape))
            elif kind == "bn_nbt":
                mod.register_buffer(parts[-1], torch.zeros((), dtype=torch.int64))
            else:
                mod.register_parameter(parts[-1], nn.Parameter(torch.zeros(shape)))
            self._names.append(rel)

    def forward(self, x):
        sd = dict(self.named_parameters())
        sd.update(dict(self.named_buffers()))
        st = O._BNState()
        y = O.effnet_block(x, sd, "", self.desc, self.training, st)
        if self.training:
            with torch.no_grad():
                for k, v in st.updates.items():
                    sd[k].copy_(v)
        return y


class _ShimBlocks(nn.Module):
    def __init__(self):
        super().__init__()
        specs = O.param_specs(O.CFG_EFF)
        stages = OrderedDict()
        for b in O.effnet_blocks():
            stages.setdefault(b["stage"], []).append(b)
        for s, blocks in stages.items():
            seq = nn.Module()
            for b in blocks:
                seq.add_module(str(b["idx"]), _ShimBlock(b, specs, f"encoder.shallow_cnn.eff_block.{s}.{b['idx']}."))
            self.add_module(str(s), seq)

    def forward(self, x):
        for stage in self.children():
            for blk in stage.children():
                x = blk(x)
        return x


def import_reference():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Model:
        def __init__(self):
            self.blocks = _ShimBlocks()

    timm = stub("timm", create_model=lambda *a, **k: _Model())
    stub("timm.models")
    stub("timm.models.layers", DropPath=nn.Identity, to_2tuple=lambda x: (x, x), trunc_normal_=lambda *a, **k: None)
    for n in ("wandb", "editdistance", "albumentations"):
        stub(n)
    stub("albumentations.pytorch", ToTensorV2=object)
    _orig = torch.Tensor.get_device
    torch.Tensor.get_device = lambda self: self.device if not self.is_cuda else _orig(self)
    sys.path.insert(0, REF)
    import utils  # noqa: F401  (first: breaks the import cycle, as train_single_opt.py:16 does)
    from networks import LiteSATRN, EfficientSATRN  # classes
    return utils, LiteSATRN, EfficientSATRN


def flags_for(utils, cfg, height, width):
    d = dict(network=cfg["network"], input_size=dict(height=height, width=width),
             SATRN=dict(encoder=dict(hidden_dim=cfg["enc_hidden"], filter_dim=cfg["enc_filter"],
                                     layer_num=cfg["enc_layers"], head_num=cfg["enc_heads"]),
                        decoder=dict(src_dim=cfg["dec_src"], hidden_dim=cfg["dec_hidden"],
                                     filter_dim=cfg["dec_filter"], layer_num=cfg["dec_layers"],
                                     head_num=cfg["dec_heads"])),
             data=dict(rgb=cfg["rgb"]), dropout_rate=0.0)
    return utils.Flags(d).get()


class _DS:
    pass


def build_reference(utils, cls, cfg, height, width, seed):
    ds = _DS()
    ds.token_to_id, ds.id_to_token = utils.load_vocab([os.path.join(REF, "configs/tokens.txt")])
    assert len(ds.id_to_token) == O.NUM_CLASSES
    model = cls(flags_for(utils, cfg, height, width), ds)
    sd = O.det_state_dict(cfg, seed)
    missing = model.load_state_dict(sd, strict=True)
    for m in model.modules():  # parity mode: every dropout off, incl. the hard-wired FFN 0.1
        if isinstance(m, nn.Dropout):
            m.p = 0.0
    return model, sd


def checksum(t):
    t = t.detach().double().flatten()
    n = t.numel()
    idx = (torch.arange(64, dtype=torch.int64) * 2654435761 % max(n, 1))
    return np.array([t.sum().item(), t.abs().sum().item()], dtype=np.float64), t[idx].numpy().astype(np.float64)


def run_case(utils, cls, cfg, name, batch, height, width, seq_len, seed, full, pad_tail=0, greedy_steps=None):
    torch.manual_seed(0)
    import random
    random.seed(0)
    model, sd = build_reference(utils, cls, cfg, height, width, seed)
    img, expected = O.det_inputs(batch, cfg["rgb"], height, width, seq_len, seed=21 + seed, pad_tail=pad_tail)
    out = {}
    meta = dict(network=cfg["network"], batch=batch, height=height, width=width, seq_len=seq_len,
                wseed=seed, iseed=21 + seed, pad_tail=pad_tail)
    meta.update({k: v for k, v in cfg.items() if k != "network"})
    # ---- training forward (teacher forced) + loss + backward
    model.train()
    logits = model(img, expected, True, 1.0)  # random.random() < 1.0 always -> TF branch
    loss = model.criterion(logits.transpose(1, 2), expected[:, 1:])
    model.zero_grad()
    loss.backward()
    out["loss"] = np.array(loss.item(), dtype=np.float64)
    if full:
        out["logits"] = logits.detach().numpy()
    s, smp = checksum(logits)
    out["logits_sum"], out["logits_samples"] = s, smp
    names = O.trainable_names(cfg)
    params = dict(model.named_parameters())
    gs, gsm = [], []
    for n in names:
        g = params[n].grad
        if g is None:
            g = torch.zeros_like(params[n])
        a, b = checksum(g)
        gs.append(a)
        gsm.append(b)
        if full:
            out["grad/" + n] = g.detach().numpy()
    out["grad_sums"] = np.stack(gs)
    out["grad_samples"] = np.stack(gsm)
    # BN running stats after one train-mode forward
    bufs = dict(model.named_buffers())
    rs = []
    for n, (shape, kind) in O.param_specs(cfg).items():
        if kind in ("bn_rm", "bn_rv"):
            rs.append(checksum(bufs[n])[0])
    out["bn_running_sums"] = np.stack(rs)
    # ---- encoder output + greedy decode in eval mode with the ORIGINAL weights/buffers
    model.load_state_dict(sd, strict=True)
    model.eval()
    with torch.no_grad():
        src = model.encoder(img)
        s, smp = checksum(src)
        out["enc_sum"], out["enc_samples"] = s, smp
        if full:
            out["enc"] = src.numpy()
        steps = greedy_steps or seq_len
        exp2 = expected[:, : steps + 1]
        glog = model(img, exp2, False, 0.0)  # [b, steps, V]
        top2 = torch.topk(glog, 2, dim=-1)
        out["greedy_ids"] = top2.indices[..., 0].numpy().astype(np.int64)
        out["greedy_margin"] = (top2.values[..., 0] - top2.values[..., 1]).numpy()
        s, smp = checksum(glog)
        out["greedy_sum"], out["greedy_samples"] = s, smp
        if full:
            out["greedy_logits"] = glog.numpy()
    out["meta_keys"] = np.array(list(meta.keys()))
    out["meta_vals"] = np.array([str(v) for v in meta.values()])
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: loss={loss.item():.6f} -> {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def main():
    utils, LiteSATRN, EfficientSATRN = import_reference()
    torch.set_num_threads(8)
    lite_small = dict(O.CFG_LITE, enc_hidden=32, enc_filter=32, enc_heads=4, dec_src=32, dec_hidden=32,
                      dec_filter=64, dec_heads=4)
    run_case(utils, LiteSATRN, lite_small, "lite_small", 2, 32, 48, 6, seed=1, full=True, pad_tail=2)
    run_case(utils, LiteSATRN, O.CFG_LITE, "lite_c1", 4, 64, 192, 32, seed=2, full=False)
    run_case(utils, LiteSATRN, O.CFG_LITE, "lite_c1_pad", 4, 64, 192, 32, seed=3, full=False, pad_tail=5)
    run_case(utils, EfficientSATRN, O.CFG_EFF, "eff_small", 2, 64, 96, 8, seed=4, full=False, pad_tail=2)
    run_case(utils, EfficientSATRN, O.CFG_EFF, "eff_c2_b2", 2, 128, 384, 128, seed=5, full=False, greedy_steps=16)


if __name__ == "__main__":
    main()
