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
    elif model_type == "SWIN":
        model = networks.SWIN(FLAGS, dataset, model_checkpoint or True, decoding_manager, **kw).to(device)
    else:
        raise NotImplementedError
    return model


def id_to_string(tokens, data_loader, do_eval=0):
    """utils/utils.py:134-164: token ids [B, T] -> list of space-joined token strings (each token followed by a space).
    do_eval: <PAD>/<SOS> skipped, decoding of a row stops at its first <EOS>; -1 entries are always skipped.
    The reference calls .item() per token (B*T host syncs on a device tensor); here the ids cross to the host ONCE."""
    rows = tokens.detach().cpu().tolist() if hasattr(tokens, "detach") else [list(r) for r in tokens]
    id_to_token = data_loader.dataset.id_to_token
    result = []
    if do_eval:
        t2i = data_loader.dataset.token_to_id
        eos_id = t2i["<EOS>"]
        special = {t2i["<PAD>"], t2i["<SOS>"], eos_id}
    for example in rows:
        parts = []
        for token in example:
            if do_eval:
                if token not in special:
                    if token != -1:
                        parts.append(id_to_token[token])
                elif token == eos_id:
                    break
            elif token != -1:
                parts.append(id_to_token[token])
        result.append("".join(p + " " for p in parts))
    return result
