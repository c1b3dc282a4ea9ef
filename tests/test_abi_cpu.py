"""CPU tests: the C-ABI library loads, exports every symbol include/satrn_hip.h declares, its state table is the
reference's state_dict layout, and the host-side mirror behaves like the reference's factory (no compute calls)."""
import os

import pytest
import torch

from oracle import satrn_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sa():
    import satrn_amd
    return satrn_amd


class _DS:
    token_to_id = {"<SOS>": 0, "<EOS>": 1, "<PAD>": 2}
    id_to_token = {i: str(i) for i in range(O.NUM_CLASSES)}


def flags(sa, cfg, h, w):
    return sa.Flags(dict(network=cfg["network"], input_size=dict(height=h, width=w),
                         SATRN=dict(encoder=dict(hidden_dim=cfg["enc_hidden"], filter_dim=cfg["enc_filter"], layer_num=cfg["enc_layers"], head_num=cfg["enc_heads"]),
                                    decoder=dict(src_dim=cfg["dec_src"], hidden_dim=cfg["dec_hidden"], filter_dim=cfg["dec_filter"], layer_num=cfg["dec_layers"], head_num=cfg["dec_heads"])),
                         data=dict(rgb=cfg["rgb"]), dropout_rate=0.1)).get()


def test_library_exports_every_declared_symbol(sa):
    sigs = sa._lib.parse_header()
    assert len(sigs) >= 45
    lib = sa._lib.load()
    for name in sigs:
        assert hasattr(lib, name), name
    assert lib.satrn_abi_version() == 1


@pytest.mark.parametrize("cfg,cls,h,w,nparam", [(O.CFG_LITE, "LiteSATRN", 64, 192, 2633077), (O.CFG_EFF, "EfficientSATRN", 128, 384, 27221141)])
def test_state_dict_layout_matches_reference(sa, cfg, cls, h, w, nparam):
    model = getattr(sa, cls)(flags(sa, cfg, h, w), _DS(), dtype="bf16")
    sd = model.state_dict()
    spec = O.param_specs(cfg)  # pinned to the reference by tests/test_oracle_golden.py (load_state_dict(strict=True))
    assert set(sd) == set(spec)
    for k, (shape, kind) in spec.items():
        assert tuple(sd[k].shape) == tuple(shape), k
    assert sum(p.numel() for p in model.parameters()) == nparam
    # reference callers use these (train_modules/train_single_opt.py:295-303)
    assert len(list(model.encoder.parameters())) + len(list(model.decoder.parameters())) == len(list(model.parameters()))
    assert model.decoder.layer_num == cfg["dec_layers"] and model.decoder.st_id == 0
    # checkpoints round-trip by key
    det = O.det_state_dict(cfg, 3)
    model.load_state_dict(det)
    for k in det:
        assert torch.equal(model.state_dict()[k], det[k]), k
    # workspace sizing is a host-only dry run of the plan
    ws = model._lib.satrn_model_workspace_bytes(model._h, 4, 33)
    assert 1 << 20 < ws < 64 << 30


def test_host_helpers_mirror_reference(sa, tmp_path):
    p = tmp_path / "tokens.txt"
    p.write_text("\n".join(f"t{i}" for i in range(241)) + "\n")
    t2i, i2t = sa.load_vocab([str(p)])
    assert len(t2i) == 245 and t2i["<SOS>"] == 0 and t2i["<EOS>"] == 1 and t2i["<PAD>"] == 2 and t2i[""] == 244
    f = sa.Flags(dict(optimizer=dict(lr="5e-4"), prefix="log/x", n="3")).get()
    assert f.optimizer.lr == 5e-4 and f.prefix == "./log/x" and f.n == 3
    with pytest.raises(NotImplementedError):   # unknown names raise like utils/utils.py:78-79 (ASTER is out of scope; SWIN is served)
        sa.get_network("ASTER", None, None, "cpu", None)


def test_product_fails_loudly_without_gpu(sa):
    model = sa.LiteSATRN(flags(sa, O.CFG_LITE, 64, 192), _DS(), None, dtype="f32")
    img, expected = O.det_inputs(2, 1, 64, 192, 8)
    with pytest.raises(sa.SatrnError):
        model(img, expected, True, 1.0)
    with pytest.raises(sa.SatrnError):
        model.criterion(torch.zeros(2, 245, 8), expected[:, 1:])


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "p4-fr-sorry-math-but-love-you_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in txt.replace("no oracle", ""), fn
