"""CPU: the oracle (oracle/satrn_oracle.py) against the golden vectors produced by the reference itself
(tests/golden/make_golden.py).  This is the pin that lets the GPU parity tests trust the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import satrn_oracle as O

CASES = ["lite_small", "lite_c1", "lite_c1_pad", "eff_small", "eff_c2_b2"]


def load_case(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    meta = dict(zip(z["meta_keys"].tolist(), z["meta_vals"].tolist()))
    cfg = dict(network=meta["network"])
    for k in ("rgb", "enc_hidden", "enc_filter", "enc_heads", "enc_layers", "dec_src", "dec_hidden",
              "dec_filter", "dec_heads", "dec_layers", "num_classes"):
        cfg[k] = int(meta[k])
    return z, meta, cfg


def checksum(t):
    t = t.detach().double().flatten()
    n = t.numel()
    idx = (torch.arange(64, dtype=torch.int64) * 2654435761 % max(n, 1))
    return np.array([t.sum().item(), t.abs().sum().item()]), t[idx].numpy()


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference(golden_dir, name):
    torch.set_num_threads(8)
    z, meta, cfg = load_case(golden_dir, name)
    B, H, W, T = (int(meta[k]) for k in ("batch", "height", "width", "seq_len"))
    sd = O.det_state_dict(cfg, int(meta["wseed"]))
    img, expected = O.det_inputs(B, cfg["rgb"], H, W, T, seed=int(meta["iseed"]), pad_tail=int(meta["pad_tail"]))
    loss, logits, grads, bn = O.forward_backward(img, expected, sd, cfg)
    assert abs(loss.item() - float(z["loss"])) < 2e-5
    s, smp = checksum(logits)
    np.testing.assert_allclose(smp, z["logits_samples"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(s[1], z["logits_sum"][1], rtol=1e-5)
    if "logits" in z:
        np.testing.assert_allclose(logits.numpy(), z["logits"], rtol=0, atol=1e-4)
    names = O.trainable_names(cfg)
    for i, n in enumerate(names):
        a, b = checksum(grads[n])
        ref_abs = z["grad_sums"][i][1]
        np.testing.assert_allclose(a[1], ref_abs, rtol=2e-3, atol=1e-6, err_msg=n)
        np.testing.assert_allclose(b, z["grad_samples"][i], rtol=0, atol=2e-5 + 2e-3 * np.abs(z["grad_samples"][i]).max(), err_msg=n)
        if "grad/" + n in z:
            g = z["grad/" + n]
            np.testing.assert_allclose(grads[n].numpy(), g, rtol=0, atol=1e-5 + 1e-3 * np.abs(g).max(), err_msg=n)
    rs = [checksum(bn[n])[0] for n, (_, kind) in O.param_specs(cfg).items() if kind in ("bn_rm", "bn_rv")]
    np.testing.assert_allclose(np.stack(rs), z["bn_running_sums"], rtol=1e-4, atol=1e-5)
    # eval: encoder output and greedy decode (token ids bit-exact where the top-1/top-2 margin is clear)
    with torch.no_grad():
        src = O.encoder_forward(img, sd, cfg, False)
        np.testing.assert_allclose(checksum(src)[1], z["enc_samples"], rtol=0, atol=2e-4)
        steps = z["greedy_ids"].shape[1]
        glog, ids = O.decoder_greedy_forward(src, steps, sd, cfg)
        clear = z["greedy_margin"] > 1e-4
        assert (ids.numpy()[clear] == z["greedy_ids"][clear]).all()
        np.testing.assert_allclose(checksum(glog)[1], z["greedy_samples"], rtol=0, atol=5e-4)
        if "greedy_logits" in z:
            np.testing.assert_allclose(glog.numpy(), z["greedy_logits"], rtol=0, atol=2e-4)
