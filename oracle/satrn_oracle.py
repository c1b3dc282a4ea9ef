"""CPU oracle for the SATRN hot path.  *** TEST INFRASTRUCTURE ONLY ***

A plain PyTorch-fp32 (CPU) restatement of the reference's EfficientSATRN / LiteSATRN
forward path, written functionally over a ``state_dict`` that uses the reference's own key
names.  Backward comes from autograd of this restatement.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file; the
product (``p4-fr-sorry-math-but-love-you_amd/``) never does and fails loudly without its HIP
library.

Parity pin: ``tests/golden/make_golden.py`` imports the reference itself (from
/root/reference, authoring container only) and stores its outputs in ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks this file against those fixtures.  The 40
EfficientNetV2-S blocks come from timm==0.4.9 (requirements.txt:16), which is NOT under
/root/reference and not installable here: for the blocks themselves parity is UNPINNED
(restated from the public architecture); everything around them is
pinned through the reference's own ``EfficientSATRN`` class with these blocks injected.

Every function cites the reference file:line it follows (paths relative to /root/reference).
All dropout is p=0 here (parity mode); BN runs in batch-stat mode when ``train`` is True.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# vocabulary constants  (utils/data_utils.py:6-9,24-42; data/dataset.py:12-15)
# --------------------------------------------------------------------------------------
SOS_ID, EOS_ID, PAD_ID = 0, 1, 2
NUM_CLASSES = 245  # 3 specials + 241 tokens + "" (configs/tokens.txt)

# --------------------------------------------------------------------------------------
# EfficientNetV2-S block table (timm==0.4.9 tf_efficientnetv2_s)
#   (type, repeats, stride_first, expand, out_ch, se_ratio)
# --------------------------------------------------------------------------------------
EFFNETV2_S = [
    ("cn", 2, 1, 1, 24, 0.0),
    ("er", 4, 2, 4, 48, 0.0),
    ("er", 4, 2, 4, 64, 0.0),
    ("ir", 6, 2, 4, 128, 0.25),
    ("ir", 9, 1, 6, 160, 0.25),
    ("ir", 15, 2, 6, 256, 0.25),
]
BN_EPS_TF = 1e-3


def effnet_blocks(stem_ch=24, table=EFFNETV2_S):
    """Flat list of block descriptors: dict(stage, idx, type, cin, cout, mid, stride, se, skip)."""
    out = []
    cin = stem_ch
    for s, (typ, rep, stride, exp, cout, se) in enumerate(table):
        for i in range(rep):
            st = stride if i == 0 else 1
            mid = cin * exp
            out.append(dict(stage=s, idx=i, type=typ, cin=cin, cout=cout, mid=mid, stride=st,
                            se=int(cin * se) if se > 0 else 0, skip=(st == 1 and cin == cout)))
            cin = cout
    return out


# --------------------------------------------------------------------------------------
# parameter specs (reference state_dict layout)
# kinds: xavier (xavier_normal_), conv (kaiming-uniform default), linear_w/linear_b (torch
# default), bn_w/bn_b/ln_w/ln_b, bn_rm/bn_rv/bn_nbt (buffers), embed (N(0,1))
# --------------------------------------------------------------------------------------
def _bn(spec, name, c):
    spec[name + ".weight"] = ((c,), "bn_w")
    spec[name + ".bias"] = ((c,), "bn_b")
    spec[name + ".running_mean"] = ((c,), "bn_rm")
    spec[name + ".running_var"] = ((c,), "bn_rv")
    spec[name + ".num_batches_tracked"] = ((), "bn_nbt")


def _mha(spec, name, qc, kc):
    # networks/EfficientSATRN.py:176-196
    spec[name + ".q_linear.weight"] = ((qc, qc), "xavier")
    spec[name + ".q_linear.bias"] = ((qc,), "linear_b")
    spec[name + ".k_linear.weight"] = ((qc, kc), "xavier")
    spec[name + ".k_linear.bias"] = ((qc,), "linear_b")
    spec[name + ".v_linear.weight"] = ((qc, kc), "xavier")
    spec[name + ".v_linear.bias"] = ((qc,), "linear_b")
    spec[name + ".out_linear.weight"] = ((qc, qc), "xavier")
    spec[name + ".out_linear.bias"] = ((qc,), "linear_b")


def param_specs(cfg):
    """cfg: dict(network, rgb, enc_hidden, enc_filter, enc_heads, enc_layers, dec_src, dec_hidden,
    dec_filter, dec_heads, dec_layers, num_classes)."""
    s = OrderedDict()
    D = cfg["enc_hidden"]
    if cfg["network"] == "LiteSATRN":
        # networks/LiteSATRN.py:21-48
        chans = [cfg["rgb"], D // 2, D, D, D]
        for i in range(4):
            s[f"encoder.shallow_cnn.conv{i}.weight"] = ((chans[i + 1], chans[i], 3, 3), "xavier" if i < 3 else "conv")
            _bn(s, f"encoder.shallow_cnn.batch_norm{i}", chans[i + 1])
    else:
        # networks/EfficientSATRN.py:63-79 + timm blocks
        p = "encoder.shallow_cnn."
        s[p + "conv_stem.weight"] = ((24, cfg["rgb"], 3, 3), "conv")
        _bn(s, p + "bn1", 24)
        for b in effnet_blocks():
            q = f"{p}eff_block.{b['stage']}.{b['idx']}."
            if b["type"] == "cn":
                s[q + "conv.weight"] = ((b["cout"], b["cin"], 3, 3), "conv")
                _bn(s, q + "bn1", b["cout"])
            elif b["type"] == "er":
                s[q + "conv_exp.weight"] = ((b["mid"], b["cin"], 3, 3), "conv")
                _bn(s, q + "bn1", b["mid"])
                s[q + "conv_pwl.weight"] = ((b["cout"], b["mid"], 1, 1), "conv")
                _bn(s, q + "bn2", b["cout"])
            else:
                s[q + "conv_pw.weight"] = ((b["mid"], b["cin"], 1, 1), "conv")
                _bn(s, q + "bn1", b["mid"])
                s[q + "conv_dw.weight"] = ((b["mid"], 1, 3, 3), "conv")
                _bn(s, q + "bn2", b["mid"])
                s[q + "se.conv_reduce.weight"] = ((b["se"], b["mid"], 1, 1), "conv")
                s[q + "se.conv_reduce.bias"] = ((b["se"],), "linear_b")
                s[q + "se.conv_expand.weight"] = ((b["mid"], b["se"], 1, 1), "conv")
                s[q + "se.conv_expand.bias"] = ((b["mid"],), "linear_b")
                s[q + "conv_pwl.weight"] = ((b["cout"], b["mid"], 1, 1), "conv")
                _bn(s, q + "bn3", b["cout"])
        s[p + "conv_last.weight"] = ((D, 256, 1, 1), "conv")
        _bn(s, p + "bn2", D)
    # networks/EfficientSATRN.py:102-109
    s["encoder.positional_encoding.dense0.weight"] = ((D // 2, D), "xavier")
    s["encoder.positional_encoding.dense0.bias"] = ((D // 2,), "linear_b")
    s["encoder.positional_encoding.dense1.weight"] = ((2 * D, D // 2), "xavier")
    s["encoder.positional_encoding.dense1.bias"] = ((2 * D,), "linear_b")
    Fe = cfg["enc_filter"]
    for l in range(cfg["enc_layers"]):
        q = f"encoder.attention_layers.{l}."
        # networks/EfficientSATRN.py:235-257
        s[q + "norm.weight"] = ((D,), "ln_w")
        s[q + "norm.bias"] = ((D,), "ln_b")
        _mha(s, q + "attention_layer", D, D)
        s[q + "conv0.weight"] = ((Fe, D, 1, 1), "xavier")
        _bn(s, q + "norm0", Fe)
        s[q + "depthwise.weight"] = ((Fe, 1, 3, 3), "xavier")
        s[q + "depthwise.bias"] = ((Fe,), "linear_b")
        _bn(s, q + "depthwise_norm", Fe)
        s[q + "conv1.weight"] = ((D, Fe, 1, 1), "xavier")
        _bn(s, q + "norm1", D)
    Dd, Ds, Ff = cfg["dec_hidden"], cfg["dec_src"], cfg["dec_filter"]
    V = cfg["num_classes"]
    s["decoder.embedding.weight"] = ((V + 1, Dd), "embed")  # :445
    for l in range(cfg["dec_layers"]):
        q = f"decoder.attention_layers.{l}."
        # networks/EfficientSATRN.py:353-372
        _mha(s, q + "self_attention_layer", Dd, Dd)
        s[q + "self_attention_norm.weight"] = ((Dd,), "ln_w")
        s[q + "self_attention_norm.bias"] = ((Dd,), "ln_b")
        _mha(s, q + "attention_layer", Dd, Ds)
        s[q + "attention_norm.weight"] = ((Dd,), "ln_w")
        s[q + "attention_norm.bias"] = ((Dd,), "ln_b")
        s[q + "feedforward_layer.linear0.weight"] = ((Ff, Dd), "xavier")
        s[q + "feedforward_layer.linear0.bias"] = ((Ff,), "linear_b")
        s[q + "feedforward_layer.linear1.weight"] = ((Dd, Ff), "xavier")
        s[q + "feedforward_layer.linear1.bias"] = ((Dd,), "linear_b")
        s[q + "feedforward_norm.weight"] = ((Dd,), "ln_w")
        s[q + "feedforward_norm.bias"] = ((Dd,), "ln_b")
    s["decoder.generator.weight"] = ((V, Dd), "linear_w")
    s["decoder.generator.bias"] = ((V,), "linear_b")
    return s


CFG_LITE = dict(network="LiteSATRN", rgb=1, enc_hidden=256, enc_filter=256, enc_heads=4, enc_layers=1,
                dec_src=256, dec_hidden=128, dec_filter=512, dec_heads=4, dec_layers=2,
                num_classes=NUM_CLASSES)  # configs/LiteSATRN.yaml:5-16
CFG_EFF = dict(network="EfficientSATRN", rgb=1, enc_hidden=512, enc_filter=512, enc_heads=8, enc_layers=2,
               dec_src=512, dec_hidden=256, dec_filter=1024, dec_heads=8, dec_layers=3,
               num_classes=NUM_CLASSES)  # configs/EfficientSATRN.yaml:5-16


# --------------------------------------------------------------------------------------
# deterministic weights / inputs: counter-based integer hash -> uniform.  Build-owned, so the
# same tensors can be regenerated on the GPU box without shipping them.
# --------------------------------------------------------------------------------------
def _hash_uniform(n, seed):
    """n float32 values in [-1, 1) from a 32-bit mix of (seed, index).  Pure integer math."""
    i = torch.arange(n, dtype=torch.int64)
    x = (i * 0x9E3779B1 + (seed + 1) * 0x85EBCA77) & 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return ((x >> 8).to(torch.float64) / float(1 << 23) - 1.0).to(torch.float32)


def det_tensor(shape, seed, scale=1.0):
    n = 1
    for d in shape:
        n *= d
    return (_hash_uniform(max(n, 1), seed)[:n] * scale).reshape(shape)


def _name_seed(name, seed):
    h = seed * 1000003
    for ch in name:
        h = (h * 131 + ord(ch)) & 0x7FFFFFFF
    return h


def det_state_dict(cfg, seed=0):
    """Deterministic, well-conditioned weights for every key of param_specs(cfg)."""
    sd = OrderedDict()
    for name, (shape, kind) in param_specs(cfg).items():
        sk = _name_seed(name, seed)
        if kind in ("xavier", "conv", "linear_w"):
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            fan_out = shape[0]
            for d in shape[2:]:
                fan_out *= d
            if kind == "xavier":
                a = math.sqrt(6.0 / (fan_in + fan_out))
            else:
                a = math.sqrt(3.0 / fan_in)
            t = det_tensor(shape, sk, a)
        elif kind == "linear_b":
            t = det_tensor(shape, sk, 0.1)
        elif kind in ("bn_w", "ln_w"):
            t = 1.0 + det_tensor(shape, sk, 0.2)
        elif kind in ("bn_b", "ln_b"):
            t = det_tensor(shape, sk, 0.1)
        elif kind == "bn_rm":
            t = det_tensor(shape, sk, 0.1)
        elif kind == "bn_rv":
            t = 1.0 + det_tensor(shape, sk, 0.3)
        elif kind == "bn_nbt":
            t = torch.zeros((), dtype=torch.int64)
        elif kind == "embed":
            t = det_tensor(shape, sk, 1.0)
        else:
            raise ValueError(kind)
        sd[name] = t
    return sd


def det_inputs(batch, rgb, height, width, seq_len, seed=21, pad_tail=0):
    """Synthetic batch: images ~U(-1.7,1.7) (unit variance), expected [B, T+1] with
    col 0 = SOS, last = EOS, rest uniform in 3..244; the last `pad_tail` columns of odd rows are
    PAD (exercises ignore_index and pad_mask)."""
    img = det_tensor((batch, rgb, height, width), seed * 7 + 1, 1.7320508)
    u = _hash_uniform(batch * (seq_len + 1), seed * 7 + 2).reshape(batch, seq_len + 1)
    ids = (3 + ((u + 1.0) * 0.5 * 242).floor().clamp(0, 241)).to(torch.int64)
    ids[:, 0] = SOS_ID
    ids[:, -1] = EOS_ID
    if pad_tail > 0:
        for b in range(1, batch, 2):
            ids[b, -pad_tail:] = PAD_ID
            ids[b, -pad_tail - 1] = EOS_ID
    return img, ids


# --------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------
class _BNState:
    """Collects running-stat updates so callers can compare them with the HIP path."""

    def __init__(self):
        self.updates = OrderedDict()


def batch_norm(x, sd, name, train, eps, bnstate=None, momentum=0.1):
    """nn.BatchNorm2d semantics (batch stats + running update in train, running stats in eval)."""
    w, b = sd[name + ".weight"], sd[name + ".bias"]
    rm, rv = sd[name + ".running_mean"], sd[name + ".running_var"]
    if train:
        rm2, rv2 = rm.detach().clone(), rv.detach().clone()
        y = F.batch_norm(x, rm2, rv2, w, b, True, momentum, eps)
        if bnstate is not None:
            bnstate.updates[name + ".running_mean"] = rm2
            bnstate.updates[name + ".running_var"] = rv2
        return y
    return F.batch_norm(x, rm, rv, w, b, False, momentum, eps)


def same_pad(x, k, s):
    """TF 'SAME' padding as timm Conv2dSame computes it (extra pixel goes bottom/right)."""
    ih, iw = x.shape[-2:]
    ph = max((math.ceil(ih / s) - 1) * s + k - ih, 0)
    pw = max((math.ceil(iw / s) - 1) * s + k - iw, 0)
    return F.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2])


def conv_same(x, w, stride, groups=1):
    k = w.shape[-1]
    if stride == 1:
        return F.conv2d(x, w, None, 1, k // 2, 1, groups)
    return F.conv2d(same_pad(x, k, stride), w, None, stride, 0, 1, groups)


def effnet_block(x, sd, q, b, train, bnstate):
    """One timm==0.4.9 EfficientNetV2-S block (third-party; parity UNPINNED, see module header)."""
    sc = x
    if b["type"] == "cn":
        x = conv_same(x, sd[q + "conv.weight"], b["stride"])
        x = F.silu(batch_norm(x, sd, q + "bn1", train, BN_EPS_TF, bnstate))
    elif b["type"] == "er":
        x = conv_same(x, sd[q + "conv_exp.weight"], b["stride"])
        x = F.silu(batch_norm(x, sd, q + "bn1", train, BN_EPS_TF, bnstate))
        x = F.conv2d(x, sd[q + "conv_pwl.weight"])
        x = batch_norm(x, sd, q + "bn2", train, BN_EPS_TF, bnstate)
    else:
        x = F.conv2d(x, sd[q + "conv_pw.weight"])
        x = F.silu(batch_norm(x, sd, q + "bn1", train, BN_EPS_TF, bnstate))
        x = conv_same(x, sd[q + "conv_dw.weight"], b["stride"], groups=b["mid"])
        x = F.silu(batch_norm(x, sd, q + "bn2", train, BN_EPS_TF, bnstate))
        g = x.mean((2, 3), keepdim=True)
        g = F.silu(F.conv2d(g, sd[q + "se.conv_reduce.weight"], sd[q + "se.conv_reduce.bias"]))
        g = torch.sigmoid(F.conv2d(g, sd[q + "se.conv_expand.weight"], sd[q + "se.conv_expand.bias"]))
        x = x * g
        x = F.conv2d(x, sd[q + "conv_pwl.weight"])
        x = batch_norm(x, sd, q + "bn3", train, BN_EPS_TF, bnstate)
    if b["skip"]:
        x = x + sc
    return x


def efficientnet_forward(x, sd, train, bnstate=None, p="encoder.shallow_cnn."):
    """networks/EfficientSATRN.py:81-87 (conv_stem pad 0 stride 2, bn eps 1e-3, SiLU; blocks;
    conv_last 1x1, bn eps 1e-5, SiLU)."""
    x = F.conv2d(x, sd[p + "conv_stem.weight"], None, 2, 0)
    x = F.silu(batch_norm(x, sd, p + "bn1", train, 1e-3, bnstate))
    for b in effnet_blocks():
        x = effnet_block(x, sd, f"{p}eff_block.{b['stage']}.{b['idx']}.", b, train, bnstate)
    x = F.conv2d(x, sd[p + "conv_last.weight"])
    x = F.silu(batch_norm(x, sd, p + "bn2", train, 1e-5, bnstate))
    return x


def shallow_cnn_forward(x, sd, train, bnstate=None, p="encoder.shallow_cnn."):
    """networks/LiteSATRN.py:50-70: 4 x [conv3x3 p1 no-bias, BN, ReLU, maxpool 2x2]."""
    for i in range(4):
        x = F.conv2d(x, sd[f"{p}conv{i}.weight"], None, 1, 1)
        x = F.relu(batch_norm(x, sd, f"{p}batch_norm{i}", train, 1e-5, bnstate))
        x = F.max_pool2d(x, 2, 2)
    return x


def pos_table_2d(length, hidden):
    """networks/EfficientSATRN.py:111-127: cat(sin, cos) (not interleaved), D/2 timescales."""
    position = torch.arange(length).float()
    nts = hidden // 2
    inc = math.log(1.0e4 / 1.0) / (torch.FloatTensor([nts]) - 1)
    inv = 1.0 * torch.exp(torch.arange(nts) * -inc)
    st = position.unsqueeze(1) * inv.unsqueeze(0)
    return torch.cat((torch.sin(st), torch.cos(st)), dim=1)  # [length, hidden]


def positional_encoding_2d(x, sd, p="encoder.positional_encoding."):
    """networks/EfficientSATRN.py:135-154."""
    b, c, h, w = x.shape
    hp = pos_table_2d(h, c).unsqueeze(1).to(x.dtype)  # [h,1,c]
    wp = pos_table_2d(w, c).unsqueeze(0).to(x.dtype)  # [1,w,c]
    g = x.mean((2, 3))
    g = F.relu(F.linear(g, sd[p + "dense0.weight"], sd[p + "dense0.bias"]))
    g = torch.sigmoid(F.linear(g, sd[p + "dense1.weight"], sd[p + "dense1.bias"]))
    g = g.reshape(-1, 2, 1, c)
    e = g[:, 0:1] * hp.unsqueeze(0) + g[:, 1:2] * wp.unsqueeze(0)  # [b,h,w,c]
    return e.permute(0, 3, 1, 2) + x


def mha(q_in, k_in, v_in, sd, p, heads, mask=None):
    """networks/EfficientSATRN.py:198-228 with :164-172.  Temperature = sqrt(heads*head_dim)
    (:187-189), mask True = -inf."""
    b, ql, kl = q_in.size(0), q_in.size(1), k_in.size(1)
    D = sd[p + ".q_linear.weight"].shape[0]
    hd = D // heads
    q = F.linear(q_in, sd[p + ".q_linear.weight"], sd[p + ".q_linear.bias"]).view(b, ql, heads, hd).transpose(1, 2)
    k = F.linear(k_in, sd[p + ".k_linear.weight"], sd[p + ".k_linear.bias"]).view(b, kl, heads, hd).transpose(1, 2)
    v = F.linear(v_in, sd[p + ".v_linear.weight"], sd[p + ".v_linear.bias"]).view(b, kl, heads, hd).transpose(1, 2)
    attn = torch.matmul(q, k.transpose(2, 3)) / float((heads * hd) ** 0.5)
    if mask is not None:
        attn = attn.masked_fill(mask.unsqueeze(1), float("-inf"))
    attn = torch.softmax(attn, dim=-1)
    out = torch.matmul(attn, v).transpose(1, 2).contiguous().view(b, ql, D)
    return F.linear(out, sd[p + ".out_linear.weight"], sd[p + ".out_linear.bias"])


def encoder_layer(x, sd, q, heads, train, bnstate=None):
    """networks/EfficientSATRN.py:259-281.  One LayerNorm used twice (:265,:268); raw reshape
    [b,hw,c] -> [b,c,h,w] (:269) is a memory reinterpretation, not a transpose."""
    b, c, h, w = x.shape
    flat = x.view(b, c, h * w).transpose(1, 2)
    nw, nb = sd[q + "norm.weight"], sd[q + "norm.bias"]
    y = F.layer_norm(flat, (c,), nw, nb)
    y = mha(y, y, y, sd, q + "attention_layer", heads)
    y = F.layer_norm(y + flat, (c,), nw, nb)
    y = y.reshape(-1, c, h, w)
    y = F.conv2d(y, sd[q + "conv0.weight"])
    y = F.relu(batch_norm(y, sd, q + "norm0", train, 1e-5, bnstate))
    y = F.conv2d(y, sd[q + "depthwise.weight"], sd[q + "depthwise.bias"], 1, 1, 1, y.shape[1])
    y = F.relu(batch_norm(y, sd, q + "depthwise_norm", train, 1e-5, bnstate))
    y = F.conv2d(y, sd[q + "conv1.weight"])
    y = F.relu(batch_norm(y, sd, q + "norm1", train, 1e-5, bnstate))
    return y + x


def encoder_forward(img, sd, cfg, train, bnstate=None):
    """networks/EfficientSATRN.py:311-323 / networks/LiteSATRN.py SATRNEncoder.forward -> [b, hw, c]."""
    if cfg["network"] == "LiteSATRN":
        x = shallow_cnn_forward(img, sd, train, bnstate)
    else:
        x = efficientnet_forward(img, sd, train, bnstate)
    x = positional_encoding_2d(x, sd)
    for l in range(cfg["enc_layers"]):
        x = encoder_layer(x, sd, f"encoder.attention_layers.{l}.", cfg["enc_heads"], train, bnstate)
    b, c, h, w = x.shape
    return x.view(b, c, h * w).transpose(1, 2)


def pos_table_1d(channels, max_len=500):
    """networks/EfficientSATRN.py:408-418: interleaved sin/cos."""
    pos = torch.arange(max_len).float().unsqueeze(1)
    i = torch.arange(channels).float().unsqueeze(0)
    rates = 1 / torch.pow(10000, (2 * (i // 2)) / channels)
    pe = pos * rates
    pe[:, 0::2] = torch.sin(pe[:, 0::2])
    pe[:, 1::2] = torch.cos(pe[:, 1::2])
    return pe


def text_embedding(ids, sd):
    """networks/EfficientSATRN.py:480-483."""
    e = F.embedding(ids, sd["decoder.embedding.weight"])
    return e * math.sqrt(e.size(2))


def feedforward(x, sd, q):
    """networks/EfficientSATRN.py:339-346: ReLU after BOTH linears."""
    x = F.relu(F.linear(x, sd[q + ".linear0.weight"], sd[q + ".linear0.bias"]))
    return F.relu(F.linear(x, sd[q + ".linear1.weight"], sd[q + ".linear1.bias"]))


def decoder_layer(tgt, tgt_prev, src, mask, sd, q, heads):
    """networks/EfficientSATRN.py:374-397.  Step mode: K/V history = cat(prev OUTPUTS, current input)."""
    Dd = tgt.shape[-1]
    kv = tgt if tgt_prev is None else torch.cat([tgt_prev, tgt], 1)
    att = mha(tgt, kv, kv, sd, q + "self_attention_layer", heads, mask)
    out = F.layer_norm(att + tgt, (Dd,), sd[q + "self_attention_norm.weight"], sd[q + "self_attention_norm.bias"])
    att = mha(out, src, src, sd, q + "attention_layer", heads)
    out = F.layer_norm(att + out, (Dd,), sd[q + "attention_norm.weight"], sd[q + "attention_norm.bias"])
    ff = feedforward(out, sd, q + "feedforward_layer")
    return F.layer_norm(ff + out, (Dd,), sd[q + "feedforward_norm.weight"], sd[q + "feedforward_norm.bias"])


def decoder_masks(text):
    """networks/EfficientSATRN.py:469-478,492: (text==PAD with column 0 cleared) | strict upper triangle."""
    pad = text == PAD_ID
    pad[:, 0] = False
    L = text.size(1)
    order = torch.triu(torch.ones(L, L), diagonal=1).bool()
    return pad.unsqueeze(1) | order.unsqueeze(0)


def decoder_tf_forward(src, text, sd, cfg):
    """networks/EfficientSATRN.py:490-495 (teacher-forced branch) -> logits [b, L, V]."""
    Dd = cfg["dec_hidden"]
    tgt = text_embedding(text, sd)
    tgt = tgt + pos_table_1d(Dd)[: text.size(1)].unsqueeze(0).to(tgt.dtype)
    mask = decoder_masks(text.clone())
    for l in range(cfg["dec_layers"]):
        tgt = decoder_layer(tgt, None, src, mask, sd, f"decoder.attention_layers.{l}.", cfg["dec_heads"])
    return F.linear(tgt, sd["decoder.generator.weight"], sd["decoder.generator.bias"])


def decoder_step(target, t, feats, src, sd, cfg):
    """EfficientSATRN_decoder.step_forward (networks/EfficientSATRN.py:932-948) == one trip of the greedy loop (:534-551):
    target [b] int64, feats = per-layer history (updated in place) -> logits [b, 1, V]."""
    b = src.size(0)
    Dd = cfg["dec_hidden"]
    tgt = text_embedding(target.view(b, 1), sd) + pos_table_1d(Dd)[t].view(1, 1, Dd)
    for l in range(cfg["dec_layers"]):
        tgt = decoder_layer(tgt, feats[l], src, None, sd, f"decoder.attention_layers.{l}.", cfg["dec_heads"])
        feats[l] = tgt if feats[l] is None else torch.cat([feats[l], tgt], 1)
    return F.linear(tgt, sd["decoder.generator.weight"], sd["decoder.generator.bias"])


def decoder_greedy_forward(src, num_steps, sd, cfg):
    """networks/EfficientSATRN.py:528-561 (no DecodingManager).  Returns (logits [b,steps,V], ids [b,steps]).
    argmax ties -> lowest index (torch.argmax)."""
    b = src.size(0)
    target = torch.full((b,), SOS_ID, dtype=torch.int64)
    feats = [None] * cfg["dec_layers"]
    outs, ids = [], []
    for t in range(num_steps):
        o = decoder_step(target, t, feats, src, sd, cfg)
        target = torch.argmax(o[:, -1, :], dim=-1)
        outs.append(o[:, 0])
        ids.append(target)
    return torch.stack(outs, 1), torch.stack(ids, 1)


def ensemble_greedy_forward(srcs, num_steps, sds, cfg):
    """utils/ensemble_utils.py:70-103 without a DecodingManager: every step, each model's step_forward logits ->
    softmax, averaged over the models, argmax of the average is every model's next input.
    Returns (averaged probabilities [b, steps, V], ids [b, steps])."""
    b = srcs[0].size(0)
    target = torch.full((b,), SOS_ID, dtype=torch.int64)
    feats = [[None] * cfg["dec_layers"] for _ in sds]
    outs, ids = [], []
    for t in range(num_steps):
        acc = None
        for m, sd in enumerate(sds):
            o = decoder_step(target, t, feats[m], srcs[m], sd, cfg)[:, 0]
            pr = F.softmax(o, dim=-1)
            acc = pr if acc is None else acc + pr
        acc = acc / len(sds)
        target = torch.argmax(acc, dim=-1)
        outs.append(acc)
        ids.append(target)
    return torch.stack(outs, 1), torch.stack(ids, 1)


def model_forward(img, expected, sd, cfg, is_train, teacher_forcing=True, bnstate=None):
    """networks/EfficientSATRN.py:697-706 / networks/LiteSATRN.py:581-590."""
    src = encoder_forward(img, sd, cfg, is_train, bnstate)
    if is_train and teacher_forcing:
        return decoder_tf_forward(src, expected[:, :-1], sd, cfg)
    return decoder_greedy_forward(src, expected.size(1) - 1, sd, cfg)[0]


def loss_fn(logits, expected):
    """networks/EfficientSATRN.py:690-692 used as train_modules/train_single_opt.py:82,86."""
    return F.cross_entropy(logits.transpose(1, 2), expected[:, 1:], ignore_index=PAD_ID)


def trainable_names(cfg):
    return [k for k, (_, kind) in param_specs(cfg).items() if not kind.startswith("bn_r") and kind != "bn_nbt"]


def forward_backward(img, expected, sd, cfg, dtype=None, teacher_forcing=True, bn_train=True):
    """One teacher-forced training forward + CE + backward on the oracle.
    Returns (loss, logits, grads{name}, bn_updates{name}).  dtype=torch.bfloat16 runs the same graph with every tensor
    in bf16 (PyTorch's own bf16 kernels): the yardstick for how much gradient noise bf16 storage costs."""
    sd = OrderedDict((k, v.clone()) for k, v in sd.items())
    if dtype is not None:
        sd = OrderedDict((k, (v.to(dtype) if v.is_floating_point() else v)) for k, v in sd.items())
        img = img.to(dtype)
    names = trainable_names(cfg)
    for n in names:
        sd[n].requires_grad_(True)
    st = _BNState()
    # teacher_forcing=False: the train-time autoregressive branch with gradients (networks/EfficientSATRN.py:496-525);
    # same arithmetic as the greedy loop, run under autograd with BN in batch-stat mode
    if bn_train:
        logits = model_forward(img, expected, sd, cfg, True, teacher_forcing, st)
    else:
        # module.eval() with gradients (BatchNorm on its running statistics; the oracle has no dropout): what
        # `model.eval(); model.decoder(model.encoder(x), text, True, L, 1.0)` computes in the reference
        src = encoder_forward(img, sd, cfg, False, st)
        logits = decoder_tf_forward(src, expected[:, :-1], sd, cfg)
    loss = loss_fn(logits.float(), expected)
    grads = torch.autograd.grad(loss, [sd[n] for n in names], allow_unused=True)
    g = OrderedDict((n, (gi if gi is not None else torch.zeros_like(sd[n])).float()) for n, gi in zip(names, grads))
    return loss.detach(), logits.detach().float(), g, st.updates


def clip_adamw_step(params, grads, m, v, step, lr, wd=1e-6, max_norm=2.0, b1=0.9, b2=0.999, eps=1e-8):
    """train_modules/train_single_opt.py:95-98: clip_grad_norm_(max_norm) then AdamW.step.
    params/grads/m/v: dict name -> tensor (updated in place).  Returns total grad norm."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for n, p in params.items():
        g = grads[n] * coef
        p.mul_(1 - lr * wd)
        m[n].mul_(b1).add_(g, alpha=1 - b1)
        v[n].mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        denom = (v[n].sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m[n], denom, value=-lr / bc1)
    return total


# ---------------------------------------------------------------------------------------------------------------
# DecodingManager (postprocessing/postprocessing.py:180-388) restated over the compiled rule table
# (table[V + 8] int32: per-token flags | limit << 8, then ids sos, eos, "", "_", "{", "}").
# ---------------------------------------------------------------------------------------------------------------
R_NEXT_UNDERBAR, R_NEXT_LBRACKET, R_NOT_UNDERBAR, R_NOT_LBRACKET, R_NOT_INITIAL = 1, 2, 4, 8, 16


def sift_blacklist(cur, series, nl, nr, table):
    """MemoryNode._look_back (:337-388) -> bool [V]: tokens forbidden at the next step."""
    V = len(table) - 8
    sos, eos, empty, under, lbr, rbr = (int(v) for v in table[V:V + 6])
    black = np.zeros(V, dtype=bool)
    black[sos] = True
    if empty >= 0:
        black[empty] = True
    if nl == nr and rbr >= 0:
        black[rbr] = True
    w = int(table[cur])
    if cur == eos:
        return black
    if cur == sos:
        black |= (np.asarray(table[:V]) & R_NOT_INITIAL) != 0
        return black
    if w & R_NEXT_UNDERBAR:
        black[:] = True
        black[under] = False
        return black
    if w & R_NEXT_LBRACKET:
        black[:] = True
        black[lbr] = False
        return black
    if w & R_NOT_UNDERBAR:
        black[under] = True
    if w & R_NOT_LBRACKET:
        black[lbr] = True
    lim = w >> 8
    if lim > 0 and series >= lim:
        black[cur] = True
    return black


def sift_new_state(batch, table):
    V = len(table) - 8
    return [[int(table[V]), 1, 0, 0] for _ in range(batch)]  # current token <SOS>, run length 1, no brackets (:308-315)


def sift(x, state, table):
    """DecodingManager.sift (:189-246): softmax of the input, blacklisted entries zeroed, argmax, MemoryNode.record
    (:317-335).  x [B, V] float; state is updated in place.  Returns (targets int64 [B], masked probabilities [B, V])."""
    V = len(table) - 8
    lbr, rbr = int(table[V + 4]), int(table[V + 5])
    probs = F.softmax(x.float(), dim=-1)
    mask = torch.from_numpy(np.stack([sift_blacklist(*st, table) for st in state]))
    probs = probs.masked_fill(mask, 0)
    targets = torch.argmax(probs, dim=-1)
    for st, t in zip(state, targets.tolist()):
        st[1] = st[1] + 1 if st[0] == t else 1
        if t == lbr:
            st[2] += 1
        elif t == rbr:
            st[3] += 1
        st[0] = t
    return targets, probs


def decoder_greedy_managed(src, num_steps, sd, cfg, table):
    """networks/EfficientSATRN.py:528-561 WITH a DecodingManager: returns (masked probabilities [b, steps, V], ids)."""
    b = src.size(0)
    target = torch.full((b,), SOS_ID, dtype=torch.int64)
    feats = [None] * cfg["dec_layers"]
    state = sift_new_state(b, table)
    outs, ids = [], []
    for t in range(num_steps):
        o = decoder_step(target, t, feats, src, sd, cfg)
        target, pr = sift(o[:, -1, :], state, table)
        outs.append(pr)
        ids.append(target)
    return torch.stack(outs, 1), torch.stack(ids, 1)


def beam_search(src, sd, cfg, beam_width=5, max_sequence=230, trace=None):
    """EfficientSATRN.beam_search (networks/EfficientSATRN.py:708-867; LiteSATRN's is the same text) for topk=1, the only
    value the reference's own output packing (:857-865) accepts.  Per image a BEST-FIRST search over a priority queue
    (not a level-synchronous beam): pop the node with the lowest score = -(sum of log-probs)/len
    (postprocessing/decoding.py:80; ties: the shorter node first, :83-84), stop at the first popped <EOS> node (:764-767) or
    after max_sequence-1 expansions (:754, num_steps += beam_width per expansion); an expansion runs one decoder step
    whose self-attention history is the layer OUTPUTS of the node's ancestors (:785-800), takes log_softmax and pushes
    the beam_width best continuations (:806-829).  No <EOS> popped -> the best node left in the queue (:834-835).  The
    utterance is read root-first INCLUDING <SOS> (:842-848), padded with <PAD> / cut to max_sequence (:857-864).
    Log-probabilities accumulate in float64 as in the reference (.item() -> Python float, :815,821).
    Returns int64 [b, max_sequence]; trace (optional list) receives per image the popped node tokens in order."""
    b = src.size(0)
    Dd = cfg["dec_hidden"]
    L = cfg["dec_layers"]
    pe = pos_table_1d(Dd)
    gw, gb = sd["decoder.generator.weight"], sd["decoder.generator.bias"]
    rows = []
    for i in range(b):
        cur_src = src[i:i + 1]
        # node = [parent, token, logp (float64), len, per-layer history of OUTPUTS or None]
        nodes = [[-1, SOS_ID, 0.0, 1, [None] * L]]
        alive = {0}
        n_exp, end = 0, -1
        popped = []
        while True:
            if n_exp * beam_width >= (max_sequence - 1) * beam_width:
                break
            n = min(alive, key=lambda k: (-(nodes[k][2] / float(nodes[k][3])), nodes[k][3], k))
            alive.discard(n)
            parent, tok, logp, ln, hist = nodes[n]
            popped.append(tok)
            if tok == EOS_ID and parent != -1:
                end = n
                break
            tgt = text_embedding(torch.tensor([[tok]], dtype=torch.int64), sd) + pe[ln - 1].view(1, 1, Dd)
            hist = list(hist)
            for l in range(L):
                tgt = decoder_layer(tgt, hist[l], cur_src, None, sd, f"decoder.attention_layers.{l}.", cfg["dec_heads"])
                hist[l] = tgt if hist[l] is None else torch.cat([hist[l], tgt], 1)
            lp = F.log_softmax(F.linear(tgt, gw, gb), dim=-1)
            vals, idx = torch.topk(lp, beam_width)
            for k in range(beam_width):
                nodes.append([n, int(idx[0, 0, k]), logp + vals[0, 0, k].item(), ln + 1, hist])
                alive.add(len(nodes) - 1)
            n_exp += 1
        if end < 0:
            end = min(alive, key=lambda k: (-(nodes[k][2] / float(nodes[k][3])), nodes[k][3], k))
        utt = []
        k = end
        while k != -1:
            utt.append(nodes[k][1])
            k = nodes[k][0]
        utt = utt[::-1]
        utt = (utt + [PAD_ID] * max(0, max_sequence - len(utt)))[:max_sequence]
        rows.append(utt)
        if trace is not None:
            trace.append(popped)
    return torch.tensor(rows, dtype=torch.int64)


def loss_fn_kd(outputs, labels, teacher_outputs, T=10, alpha=0.1):
    """train_modules/train_distillation.py:49-55: outputs / teacher_outputs [B, V, T_len] logits, labels [B, T_len].
    KL(softmax(teacher/T) || softmax(student/T)) summed over everything / B (reduction="batchmean") * alpha*T^2 plus
    (1-alpha) * cross-entropy over ALL positions (no ignore_index: PAD labels count as an ordinary class)."""
    kd = F.kl_div(F.log_softmax(outputs / T, dim=1), F.softmax(teacher_outputs / T, dim=1), reduction="batchmean")
    return kd * (alpha * T * T) + F.cross_entropy(outputs, labels) * (1.0 - alpha)


# ---------------------------------------------------------------------------------------------------------------
# Per-step training metrics (train_modules/train_single_opt.py:101-109, utils/utils.py:134-164, utils/metrics.py:9-34)
# restated over token ids.  id_to_string(do_eval=1) drops <PAD>/<SOS>/-1, stops at <EOS> and joins "tok " pieces, so the
# string ends in a space and .split(" ") yields the tokens plus one trailing ''; the "" token (last vocabulary entry)
# is that same empty string.  editdistance (third-party, pinned editdistance==0.5.3, absent here) = Levenshtein distance
# between the two token lists: UNPINNED against the package, pinned to the published algorithm; the string / sentence /
# symbol parts are pinned to the reference's own functions (tests/golden/metrics.npz).
# ---------------------------------------------------------------------------------------------------------------
def metric_tokens(row, pad_id=PAD_ID, sos_id=SOS_ID, eos_id=1, empty_id=NUM_CLASSES - 1):
    """ids the reference's string would split into (the '' token and the trailing '' both become -2)."""
    out = []
    for t in row:
        t = int(t)
        if t in (pad_id, sos_id, eos_id):
            if t == eos_id:
                break
            continue
        if t != -1:
            out.append(-2 if t == empty_id else t)
    out.append(-2)
    return out


def levenshtein(a, b):
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def step_metrics(sequence, expected, pad_id=PAD_ID):
    """-> dict(sum_wer, sentences, correct_sentences, correct_symbols, total_symbols) for one batch: what one trip of the
    reference's training loop adds to its running sums (wer and sent_acc there are per-batch means: divide by sentences)."""
    B = sequence.shape[0]
    sum_wer, ok = 0.0, 0
    for b in range(B):
        p, g = metric_tokens(sequence[b].tolist()), metric_tokens(expected[b].tolist())
        sum_wer += levenshtein(p, g) / max(len(p), len(g))
        ok += int(p == g)
    exp = expected[:, 1:].clone()
    exp[exp == pad_id] = -1
    return dict(sum_wer=sum_wer, sentences=B, correct_sentences=ok, correct_symbols=int((sequence == exp).sum().item()),
                total_symbols=int((exp != -1).sum().item()))
