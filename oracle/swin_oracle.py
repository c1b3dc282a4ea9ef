"""TEST INFRASTRUCTURE (oracle): CPU restatement (plain PyTorch fp32, functional over a reference-keyed state_dict) of the
SwinTRN path of BASELINE configs[3] -- networks/SWIN.py: SwinTransformer encoder (:590-739) + TransformerDecoder (:922-1021,
the same decoder arithmetic as the SATRN models: oracle/satrn_oracle.py's decoder functions are reused with SWIN.yaml's dims).

PINNED: tests/golden/make_golden_swin.py imports the reference's own SwinTransformer and TransformerDecoder classes on CPU
(its only third-party needs, timm.models.layers.{DropPath, to_2tuple, trunc_normal_}, are three-line helpers supplied by the
harness), loads build-owned deterministic weights and stores logits / loss / gradient checksums / encoder output / greedy ids in
tests/golden/swin_*.npz; tests/test_oracle_golden_swin.py checks this restatement against them.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import satrn_oracle as O

# the reference's hard-coded geometry (networks/SWIN.py:1028-1031) and two small ones for the parity tests
SWIN_B384 = dict(img_size=384, patch_size=4, in_chans=3, embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32), window_size=12,
                 head_classes=21841)
SWIN_TINY = dict(img_size=96, patch_size=4, in_chans=3, embed_dim=32, depths=(2, 2, 2, 2), num_heads=(1, 2, 4, 8), window_size=6, head_classes=10)
SWIN_MID = dict(img_size=192, patch_size=4, in_chans=3, embed_dim=64, depths=(2, 2, 2, 2), num_heads=(2, 4, 8, 16), window_size=12, head_classes=10)
DEC_YAML = dict(dec_src=1024, dec_hidden=512, dec_filter=512, dec_heads=8, dec_layers=4)      # configs/SWIN.yaml:11-16
DEC_TINY = dict(dec_src=256, dec_hidden=64, dec_filter=64, dec_heads=4, dec_layers=2)
DEC_MID = dict(dec_src=512, dec_hidden=128, dec_filter=128, dec_heads=4, dec_layers=2)


def stage_geometry(scfg):
    """per stage: (dim, resolution, heads, [(window, shift) per block]) -- networks/SWIN.py:253-256,477-486,674-683"""
    out = []
    res = scfg["img_size"] // scfg["patch_size"]
    for i in range(4):
        blocks = []
        for j in range(scfg["depths"][i]):
            ws, shift = scfg["window_size"], (0 if j % 2 == 0 else scfg["window_size"] // 2)
            if res <= ws:
                ws, shift = res, 0
            blocks.append((ws, shift))
        out.append((scfg["embed_dim"] << i, res, scfg["num_heads"][i], blocks))
        res //= 2
    return out


def decoder_cfg(dcfg):
    return dict(network="SWIN", num_classes=O.NUM_CLASSES, **dcfg)


def param_specs(scfg, dcfg):
    """state_dict layout of networks/SWIN.py's SWIN module in torch order (own parameters, own buffers, children)."""
    s = OrderedDict()
    E, P, Cin = scfg["embed_dim"], scfg["patch_size"], scfg["in_chans"]
    R0 = scfg["img_size"] // P
    p = "encoder."
    s[p + "absolute_pos_embed"] = ((1, R0 * R0, E), "table")                       # :660-664 (a parameter of the top module: first)
    s[p + "patch_embed.proj.weight"] = ((E, Cin, P, P), "conv")                    # :559-561
    s[p + "patch_embed.proj.bias"] = ((E,), "linear_b")
    s[p + "patch_embed.norm.weight"] = ((E,), "ln_w")
    s[p + "patch_embed.norm.bias"] = ((E,), "ln_b")
    for i, (C, res, heads, blocks) in enumerate(stage_geometry(scfg)):
        for j, (ws, shift) in enumerate(blocks):
            q = f"{p}layers.{i}.blocks.{j}."
            N = ws * ws
            if shift > 0:
                s[q + "attn_mask"] = (((res // ws) ** 2, N, N), "attn_mask")           # :311
            s[q + "norm1.weight"] = ((C,), "ln_w")
            s[q + "norm1.bias"] = ((C,), "ln_b")
            s[q + "attn.relative_position_bias_table"] = (((2 * ws - 1) ** 2, heads), "table")   # :116-118
            s[q + "attn.relative_position_index"] = ((N, N), "rel_index")             # :135
            s[q + "attn.qkv.weight"] = ((3 * C, C), "linear_w")
            s[q + "attn.qkv.bias"] = ((3 * C,), "linear_b")
            s[q + "attn.proj.weight"] = ((C, C), "linear_w")
            s[q + "attn.proj.bias"] = ((C,), "linear_b")
            s[q + "norm2.weight"] = ((C,), "ln_w")
            s[q + "norm2.bias"] = ((C,), "ln_b")
            s[q + "mlp.fc1.weight"] = ((4 * C, C), "linear_w")
            s[q + "mlp.fc1.bias"] = ((4 * C,), "linear_b")
            s[q + "mlp.fc2.weight"] = ((C, 4 * C), "linear_w")
            s[q + "mlp.fc2.bias"] = ((C,), "linear_b")
        if i < 3:
            q = f"{p}layers.{i}.downsample."
            s[q + "reduction.weight"] = ((2 * C, 4 * C), "linear_w")                   # :398 (no bias)
            s[q + "norm.weight"] = ((4 * C,), "ln_w")
            s[q + "norm.bias"] = ((4 * C,), "ln_b")
    s[p + "norm.weight"] = ((8 * E,), "ln_w")
    s[p + "norm.bias"] = ((8 * E,), "ln_b")
    s[p + "head.weight"] = ((scfg["head_classes"], 8 * E), "linear_w")               # :697-701: built, never applied (:737-739)
    s[p + "head.bias"] = ((scfg["head_classes"],), "linear_b")
    full = dict(O.CFG_LITE, **decoder_cfg(dcfg))
    for k, v in O.param_specs(full).items():
        if k.startswith("decoder."):
            # SWIN.py's Feedforward is an nn.Sequential (:826-838): its two Linear layers are `layers.0` / `layers.3`, not the
            # `linear0` / `linear1` of the SATRN files; no xavier initialisation in this file (:776-793)
            k = k.replace("feedforward_layer.linear0.", "feedforward_layer.layers.0.").replace("feedforward_layer.linear1.", "feedforward_layer.layers.3.")
            s[k] = (v[0], "linear_w" if v[1] == "xavier" else v[1])
    return s


def _dec_view(sd):
    """the same tensors under the key names oracle.satrn_oracle's decoder functions use"""
    out = dict(sd)
    for k, v in sd.items():
        if "feedforward_layer.layers." in k:
            out[k.replace("feedforward_layer.layers.0.", "feedforward_layer.linear0.").replace("feedforward_layer.layers.3.", "feedforward_layer.linear1.")] = v
    return out


def rel_index(ws):
    """networks/SWIN.py:120-135"""
    co = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (co[:, :, None] - co[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def window_partition(x, ws):
    """networks/SWIN.py:49-62"""
    B, H, W, C = x.shape
    return x.view(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C)


def window_reverse(w, ws, H, W):
    """networks/SWIN.py:65-80"""
    B = int(w.shape[0] / (H * W / ws / ws))
    return w.view(B, H // ws, W // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


def attn_mask(res, ws, shift):
    """networks/SWIN.py:288-309"""
    img = torch.zeros(1, res, res, 1)
    cnt = 0
    for h in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for w in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, h, w, :] = cnt
            cnt += 1
    mw = window_partition(img, ws).view(-1, ws * ws)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return am.masked_fill(am != 0, float(-100.0)).masked_fill(am == 0, float(0.0))


def trainable_names(scfg, dcfg):
    return [k for k, (_, kind) in param_specs(scfg, dcfg).items() if kind not in ("rel_index", "attn_mask")]


def det_state_dict(scfg, dcfg, seed=0):
    sd = OrderedDict()
    for name, (shape, kind) in param_specs(scfg, dcfg).items():
        sk = O._name_seed(name, seed)
        if kind == "rel_index":
            t = rel_index(int(round(math.sqrt(shape[0]))))
        elif kind == "attn_mask":
            ws = int(round(math.sqrt(shape[1])))
            t = attn_mask(ws * int(round(math.sqrt(shape[0]))), ws, ws // 2)
        elif kind == "table":
            t = O.det_tensor(shape, sk, 0.5)
        elif kind in ("xavier", "conv", "linear_w"):
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            fan_out = shape[0]
            a = math.sqrt(6.0 / (fan_in + fan_out)) if kind == "xavier" else math.sqrt(3.0 / fan_in)
            t = O.det_tensor(shape, sk, a)
        elif kind == "linear_b":
            t = O.det_tensor(shape, sk, 0.1)
        elif kind == "ln_w":
            t = 1.0 + O.det_tensor(shape, sk, 0.2)
        elif kind == "ln_b":
            t = O.det_tensor(shape, sk, 0.1)
        elif kind == "embed":
            t = O.det_tensor(shape, sk, 1.0)
        else:
            raise ValueError(kind)
        sd[name] = t
    return sd


def window_attention(x, sd, q, heads, ws, mask):
    """WindowAttention.forward, networks/SWIN.py:152-190 (attention / projection dropout are 0 in the reference's configuration)"""
    B_, N, C = x.shape
    qkv = F.linear(x, sd[q + "qkv.weight"], sd[q + "qkv.bias"]).reshape(B_, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    qq, kk, vv = qkv[0], qkv[1], qkv[2]
    qq = qq * ((C // heads) ** -0.5)
    attn = qq @ kk.transpose(-2, -1)
    bias = sd[q + "relative_position_bias_table"][rel_index(ws).view(-1)].view(N, N, -1).permute(2, 0, 1).contiguous()
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = attn.view(B_ // nW, nW, heads, N, N) + mask.unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, heads, N, N)
    attn = torch.softmax(attn, dim=-1)
    x = (attn @ vv).transpose(1, 2).reshape(B_, N, C)
    return F.linear(x, sd[q + "proj.weight"], sd[q + "proj.bias"])


def swin_block(x, sd, q, C, res, heads, ws, shift):
    """SwinTransformerBlock.forward, networks/SWIN.py:313-376 with drop_path = identity (eval, or rate 0)"""
    B, L, _ = x.shape
    shortcut = x
    x = F.layer_norm(x, (C,), sd[q + "norm1.weight"], sd[q + "norm1.bias"]).view(B, res, res, C)
    if shift > 0:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    xw = window_partition(x, ws).view(-1, ws * ws, C)
    aw = window_attention(xw, sd, q + "attn.", heads, ws, attn_mask(res, ws, shift) if shift > 0 else None)
    x = window_reverse(aw.view(-1, ws, ws, C), ws, res, res)
    if shift > 0:
        x = torch.roll(x, shifts=(shift, shift), dims=(1, 2))
    x = shortcut + x.view(B, L, C)
    y = F.layer_norm(x, (C,), sd[q + "norm2.weight"], sd[q + "norm2.bias"])
    y = F.gelu(F.linear(y, sd[q + "mlp.fc1.weight"], sd[q + "mlp.fc1.bias"]))          # nn.GELU: exact erf form (:24-47)
    return x + F.linear(y, sd[q + "mlp.fc2.weight"], sd[q + "mlp.fc2.bias"])


def patch_merging(x, sd, q, C, res):
    """PatchMerging.forward, networks/SWIN.py:400-422"""
    B = x.shape[0]
    x = x.view(B, res, res, C)
    x = torch.cat([x[:, 0::2, 0::2, :], x[:, 1::2, 0::2, :], x[:, 0::2, 1::2, :], x[:, 1::2, 1::2, :]], -1).view(B, -1, 4 * C)
    x = F.layer_norm(x, (4 * C,), sd[q + "norm.weight"], sd[q + "norm.bias"])
    return F.linear(x, sd[q + "reduction.weight"])


def encoder_forward(img, sd, scfg):
    """SwinTransformer.forward_features, networks/SWIN.py:722-735 (ape on, pos_drop p = 0) -> [B, (res/8)^2, 8*embed]"""
    p = "encoder."
    P, E = scfg["patch_size"], scfg["embed_dim"]
    x = F.conv2d(img, sd[p + "patch_embed.proj.weight"], sd[p + "patch_embed.proj.bias"], stride=P).flatten(2).transpose(1, 2)
    x = F.layer_norm(x, (E,), sd[p + "patch_embed.norm.weight"], sd[p + "patch_embed.norm.bias"])
    x = x + sd[p + "absolute_pos_embed"]
    for i, (C, res, heads, blocks) in enumerate(stage_geometry(scfg)):
        for j, (ws, shift) in enumerate(blocks):
            x = swin_block(x, sd, f"{p}layers.{i}.blocks.{j}.", C, res, heads, ws, shift)
        if i < 3:
            x = patch_merging(x, sd, f"{p}layers.{i}.downsample.", C, res)
    return F.layer_norm(x, (8 * E,), sd[p + "norm.weight"], sd[p + "norm.bias"])


def forward_backward(img, expected, sd, scfg, dcfg):
    """teacher-forced training forward + CE + backward (SWIN.forward, networks/SWIN.py:1056-1065; loss :1049-1051)"""
    sd = OrderedDict((k, v.clone()) for k, v in sd.items())
    names = trainable_names(scfg, dcfg)
    for n in names:
        sd[n].requires_grad_(True)
    cfg = decoder_cfg(dcfg)
    src = encoder_forward(img, sd, scfg)
    logits = O.decoder_tf_forward(src, expected[:, :-1], _dec_view(sd), cfg)
    loss = O.loss_fn(logits, expected)
    grads = torch.autograd.grad(loss, [sd[n] for n in names], allow_unused=True)
    g = OrderedDict((n, (gi if gi is not None else torch.zeros_like(sd[n])).float()) for n, gi in zip(names, grads))
    return loss.detach(), logits.detach(), g, src.detach()


def greedy(img, num_steps, sd, scfg, dcfg):
    with torch.no_grad():
        src = encoder_forward(img, sd, scfg)
        return O.decoder_greedy_forward(src, num_steps, _dec_view(sd), decoder_cfg(dcfg))
