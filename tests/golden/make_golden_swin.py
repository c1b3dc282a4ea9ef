"""Generate the SwinTRN golden fixtures by running THE REFERENCE's own classes on CPU (authoring container only):
    python tests/golden/make_golden_swin.py

networks/SWIN.py's `SWIN` module downloads ImageNet weights in its constructor (:1033, no network here), so -- as SURVEY.md
Appendix C says -- its two halves, `SwinTransformer` and `TransformerDecoder`, are instantiated directly and composed exactly
as SWIN.forward does (:1056-1065).  timm is absent: the three helpers SWIN.py imports from timm.models.layers (DropPath,
to_2tuple, trunc_normal_) are supplied by this harness (they are not on the arithmetic path: drop_path_rate is 0 here and the
weights are loaded).  Build-owned deterministic weights (oracle.swin_oracle.det_state_dict) are loaded by key; only the
resulting vectors (data) are committed."""
import os
import random
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
from oracle import satrn_oracle as O  # noqa: E402
from oracle import swin_oracle as SO  # noqa: E402

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class DropPath(nn.Module):  # timm.models.layers.DropPath: per-sample stochastic depth (identity in eval / p = 0)
        def __init__(self, p=0.0):
            super().__init__()
            self.p = p

        def forward(self, x):
            if self.p == 0.0 or not self.training:
                return x
            keep = 1 - self.p
            mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
            return x * mask / keep

    stub("timm", create_model=lambda *a, **k: None)
    stub("timm.models")
    stub("timm.models.layers", DropPath=DropPath, to_2tuple=lambda x: (x, x), trunc_normal_=lambda t, std=1.0: t)
    for n in ("wandb", "editdistance", "albumentations"):
        stub(n)
    stub("albumentations.pytorch", ToTensorV2=object)
    _orig = torch.Tensor.get_device
    torch.Tensor.get_device = lambda self: self.device if not self.is_cuda else _orig(self)
    sys.path.insert(0, REF)
    import utils  # noqa: F401
    import importlib
    return importlib.import_module("networks.SWIN")


class RefSwin(nn.Module):
    """encoder + decoder composed as networks/SWIN.py:1024-1065 does"""

    def __init__(self, S, scfg, dcfg):
        super().__init__()
        self.encoder = S.SwinTransformer(img_size=scfg["img_size"], patch_size=scfg["patch_size"], in_chans=scfg["in_chans"],
                                         embed_dim=scfg["embed_dim"], depths=list(scfg["depths"]), num_heads=list(scfg["num_heads"]),
                                         window_size=scfg["window_size"], mlp_ratio=4.0, num_classes=scfg["head_classes"],
                                         drop_path_rate=0.0, ape=True)
        self.decoder = S.TransformerDecoder(num_classes=O.NUM_CLASSES, src_dim=dcfg["dec_src"], hidden_dim=dcfg["dec_hidden"],
                                            filter_dim=dcfg["dec_filter"], head_num=dcfg["dec_heads"], dropout_rate=0.0,
                                            pad_id=O.PAD_ID, st_id=O.SOS_ID, layer_num=dcfg["dec_layers"])
        self.criterion = nn.CrossEntropyLoss(ignore_index=O.PAD_ID)

    def forward(self, input, expected, is_train, teacher_forcing_ratio):
        return self.decoder(self.encoder(input), expected[:, :-1], is_train, expected.size(1), teacher_forcing_ratio)


def checksum(t):
    t = t.detach().double().flatten()
    n = t.numel()
    idx = (torch.arange(64, dtype=torch.int64) * 2654435761 % max(n, 1))
    return np.array([t.sum().item(), t.abs().sum().item()], dtype=np.float64), t[idx].numpy().astype(np.float64)


def run_case(S, name, scfg, dcfg, batch, seq_len, seed, full, pad_tail=0, greedy_steps=6):
    torch.manual_seed(0)
    random.seed(0)
    model = RefSwin(S, scfg, dcfg)
    sd = SO.det_state_dict(scfg, dcfg, seed)
    model.load_state_dict(sd, strict=True)   # every key of the reference module, buffers included
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
    img, expected = O.det_inputs(batch, 3, scfg["img_size"], scfg["img_size"], seq_len, seed=21 + seed, pad_tail=pad_tail)
    out = {}
    model.train()
    logits = model(img, expected, True, 1.0)
    loss = model.criterion(logits.transpose(1, 2), expected[:, 1:])
    model.zero_grad()
    loss.backward()
    out["loss"] = np.array(loss.item(), dtype=np.float64)
    out["logits_sum"], out["logits_samples"] = checksum(logits)
    if full:
        out["logits"] = logits.detach().numpy()
    params = dict(model.named_parameters())
    gs, gsm = [], []
    for n in SO.trainable_names(scfg, dcfg):
        g = params[n].grad if params[n].grad is not None else torch.zeros_like(params[n])
        a, b = checksum(g)
        gs.append(a)
        gsm.append(b)
        if full and g.numel() <= 4096:
            out["grad/" + n] = g.detach().numpy()
    out["grad_sums"], out["grad_samples"] = np.stack(gs), np.stack(gsm)
    model.eval()
    with torch.no_grad():
        src = model.encoder(img)
        out["enc_sum"], out["enc_samples"] = checksum(src)
        glog = model(img, expected[:, : greedy_steps + 1], False, 0.0)
        top2 = torch.topk(glog, 2, dim=-1)
        out["greedy_ids"] = top2.indices[..., 0].numpy().astype(np.int64)
        out["greedy_margin"] = (top2.values[..., 0] - top2.values[..., 1]).numpy()
        out["greedy_sum"], out["greedy_samples"] = checksum(glog)
    meta = dict(batch=batch, seq_len=seq_len, wseed=seed, iseed=21 + seed, pad_tail=pad_tail, greedy_steps=greedy_steps)
    out["meta_keys"] = np.array(list(meta.keys()))
    out["meta_vals"] = np.array([str(v) for v in meta.values()])
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: loss={loss.item():.6f} -> {path} ({os.path.getsize(path) / 1024:.1f} KiB)", flush=True)


def main():
    S = import_reference()
    torch.set_num_threads(8)
    run_case(S, "swin_tiny", SO.SWIN_TINY, SO.DEC_TINY, 2, 6, seed=11, full=True, pad_tail=2)
    run_case(S, "swin_mid", SO.SWIN_MID, SO.DEC_MID, 2, 12, seed=12, full=False, pad_tail=3)
    if "--full" in sys.argv:   # the reference's own geometry (Swin-B / 384, SWIN.yaml decoder): ~110 M parameters, one image
        run_case(S, "swin_b384_b2", SO.SWIN_B384, SO.DEC_YAML, 2, 8, seed=13, full=False, greedy_steps=4)  # B >= 2: the reference greedy loop breaks on target.squeeze() for one image (:1013)


if __name__ == "__main__":
    main()
