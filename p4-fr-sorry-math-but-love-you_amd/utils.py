"""Host-side mirror of the reference factory/config helpers the hot path touches.
Reference: utils/utils.py:29-80 (get_network), utils/flags.py:9-45 (Flags), utils/data_utils.py:6-42 (vocab)."""
import collections
import os

import yaml

START = "<SOS>"
END = "<EOS>"
PAD = "<PAD>"
SPECIAL_TOKENS = [START, END, PAD]


def load_vocab(tokens_paths):
    """utils/data_utils.py:24-42: 3 specials + the lines of tokens.txt (incl. the trailing "") -> 245 ids."""
    tokens = list(SPECIAL_TOKENS)
    for path in tokens_paths:
        with open(path, "r") as fd:
            for tok in fd.read().split("\n"):
                if tok not in tokens:
                    tokens.append(tok)
    return {t: i for i, t in enumerate(tokens)}, {i: t for i, t in enumerate(tokens)}


def _to_namedtuple(d):
    d = dict(d)
    tup = collections.namedtuple("FLAGS", sorted(d.keys()))
    for k, v in d.items():
        if k == "prefix":
            v = os.path.join("./", v)
            d[k] = v
        if isinstance(v, dict):
            d[k] = _to_namedtuple(v)
        elif isinstance(v, str):
            # the reference eval()s strings (utils/flags.py:23) so that "5e-4" becomes a float; do that safely
            try:
                d[k] = float(v) if any(c in v for c in ".eE") else int(v)
            except ValueError:
                d[k] = v
    return tup(**d)


class Flags:
    """utils/flags.py:32-45: YAML path or dict -> nested namedtuple."""

    def __init__(self, config_file):
        if isinstance(config_file, dict):
            d = config_file
        else:
            with open(config_file, "r") as f:
                d = yaml.safe_load(f)
        self.flags = _to_namedtuple(d)

    def get(self):
        return self.flags


def get_network(model_type, FLAGS, model_checkpoint, device, dataset, decoding_manager=None, dtype=None):
    """utils/utils.py:29-80 for the model types on the accelerated path; unknown names raise NotImplementedError
    exactly like the reference (:78-79)."""
    from . import networks
    kw = {} if dtype is None else {"dtype": dtype}
    if model_type in ("EfficientSATRN", "MySATRN"):
        model = networks.EfficientSATRN(FLAGS, dataset, model_checkpoint, decoding_manager, **kw).to(device)
    elif model_type == "LiteSATRN":
        model = networks.LiteSATRN(FLAGS, dataset, model_checkpoint, decoding_manager, **kw).to(device)
    elif model_type in ("EfficientSATRN_encoder", "MySATRN_encoder"):
        model = networks.EfficientSATRN_encoder(FLAGS, dataset, model_checkpoint, **kw).to(device)
    elif model_type in ("EfficientSATRN_decoder", "MySATRN_decoder"):
        model = networks.EfficientSATRN_decoder(FLAGS, dataset, model_checkpoint, **kw).to(device)
    else:
        raise NotImplementedError
    return model
