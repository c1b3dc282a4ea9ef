"""Beam-search fixture cases shared by the generator (make_golden_beam.py, reference side) and the tests (data only)."""
from oracle import satrn_oracle as O

LITE_SMALL = dict(O.CFG_LITE, enc_hidden=32, enc_filter=32, enc_heads=4, dec_src=32, dec_hidden=32, dec_filter=64, dec_heads=4)
# name -> (network, cfg, batch, H, W, weight seed, beam_width, max_sequence, gen_scale, eos_like, eos_bias)
#   gen_scale: generator.weight multiplied by it (the deterministic weights alone give near-uniform distributions, i.e. a
#              breadth-first search that never goes deep); eos_like >= 0: the generator's <EOS> row is a copy of that
#              token's row, eos_bias added to its bias -- <EOS> then wins wherever that token would have been chosen.
CASES = dict(
    lite_small_flat=("lite", LITE_SMALL, 3, 32, 48, 11, 3, 12, 1.0, -1, 0.0),
    lite_small_first_eos=("lite", LITE_SMALL, 3, 32, 48, 11, 5, 20, 1.0, -1, 3.0),
    lite_small_deep=("lite", LITE_SMALL, 3, 32, 48, 11, 3, 16, 8.0, -1, 0.0),
    lite_small_eos_mid=("lite", LITE_SMALL, 3, 32, 48, 11, 5, 24, 8.0, 10, 0.5),
    lite_c1_flat=("lite", O.CFG_LITE, 2, 64, 192, 12, 5, 24, 1.0, -1, 0.0),
    lite_c1_deep=("lite", O.CFG_LITE, 2, 64, 192, 12, 4, 32, 8.0, -1, 0.0),
    eff_deep=("eff", O.CFG_EFF, 2, 64, 96, 13, 5, 16, 8.0, -1, 0.0),
)


def weights(cfg, seed, gen_scale=1.0, eos_like=-1, eos_bias=0.0):
    sd = O.det_state_dict(cfg, seed)
    w = sd["decoder.generator.weight"].clone() * gen_scale
    b = sd["decoder.generator.bias"].clone()
    if eos_like >= 0:
        w[O.EOS_ID] = w[eos_like]
        b[O.EOS_ID] = b[eos_like]
    b[O.EOS_ID] += eos_bias
    sd["decoder.generator.weight"], sd["decoder.generator.bias"] = w, b
    return sd
