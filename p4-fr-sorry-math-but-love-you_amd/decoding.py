"""DecodingManager on the device.

The reference's DecodingManager (postprocessing/postprocessing.py:180-290) keeps one Python MemoryNode per sample
(:293-388) and, every decode step, builds a blacklist per sample on the host, masks the softmax, takes the argmax and
calls .item() per sample.  The rules are a finite-state function of (last token, run length, bracket balance), so here
they are compiled ONCE into an int32 table and evaluated inside the decode kernels; the state is four ints per sample
in device memory.  The rule tables themselves are not part of this package: they are read from the manager object the
caller passes (its .rules / .tokens), exactly as the reference passes it to the model constructor.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import SatrnError, check, ptr

# per-token flag bits of the compiled table; the run-length limit sits in bits 8..
F_NEXT_UNDERBAR, F_NEXT_LBRACKET, F_NOT_UNDERBAR, F_NOT_LBRACKET, F_NOT_INITIAL = 1, 2, 4, 8, 16
N_IDS = 8  # sos, eos, empty-string, "_", "{", "}", reserved, reserved


def compile_rules(manager):
    """(.tokens, .rules) of a reference DecodingManager -> int32 array [V + 8]: V table words, then the special ids.

    Meaning of the rule lists as MemoryNode._look_back applies them (postprocessing/postprocessing.py:337-388):
    cannot_initial (after <SOS>), next_underbar / next_lbracket (everything else forbidden), cannot_next_underbar /
    cannot_next_lbracket, limit_series + limit_params (max run length of a token)."""
    tokens = list(manager.tokens)
    rules = manager.rules
    V = len(tokens)
    tid = {t: i for i, t in enumerate(tokens)}
    table = np.zeros(V + N_IDS, dtype=np.int32)

    def mark(name, bit):
        for t in rules.get(name, []) or []:
            if t in tid:
                table[tid[t]] |= bit

    mark("next_underbar", F_NEXT_UNDERBAR)
    mark("next_lbracket", F_NEXT_LBRACKET)
    mark("cannot_next_underbar", F_NOT_UNDERBAR)
    mark("cannot_next_lbracket", F_NOT_LBRACKET)
    mark("cannot_initial", F_NOT_INITIAL)
    series, params = rules.get("limit_series", {}), rules.get("limit_params", {})
    for t, on in series.items():
        if on and t in tid:
            lim = int(params[t])
            if not 0 < lim < (1 << 20):
                raise ValueError(f"limit_params[{t!r}] = {lim} out of range")
            table[tid[t]] |= lim << 8
    ids = [tid["<SOS>"], tid["<EOS>"], tid.get("", -1), tid.get("_", -1), tid.get("{", -1), tid.get("}", -1), 0, 0]
    table[V:] = np.asarray(ids, dtype=np.int32)
    return table


class DeviceDecodingManager:
    """Same interface as the reference's DecodingManager -- reset(sequence_length), sift(probs_step) ->
    (targets [B], masked probabilities [B, V] or [B, 1, V]) -- with the per-sample state and the rule evaluation on
    the device: no host synchronisation per step.  Built from a reference manager (or anything with .tokens/.rules)."""

    def __init__(self, manager, device="cuda"):
        self.tokens = list(manager.tokens)
        self.rules = manager.rules
        self.vocab_size = len(self.tokens)
        self.batch_size = int(getattr(manager, "batch_size", 0) or 0)
        self.sequence_length = None
        self._table_host = compile_rules(manager)
        self._table = None
        self._state = None
        self._device = torch.device(device)
        self._lib = _lib.load()

    @classmethod
    def wrap(cls, manager, device="cuda"):
        return manager if isinstance(manager, cls) else cls(manager, device)

    @property
    def sos_id(self):
        return int(self._table_host[self.vocab_size])

    def table(self, device=None):
        device = torch.device(device) if device is not None else self._device
        if device.type != "cuda":
            raise SatrnError("DeviceDecodingManager runs on MI355X only (no CPU fallback)")
        if self._table is None or self._table.device != device:
            self._table = torch.from_numpy(self._table_host).to(device)
            self._device = device
        return self._table

    def reset(self, sequence_length=None):
        """postprocessing.py:248-255 (the reference's models also call reset() with no argument, :563-564 of
        networks/EfficientSATRN.py, which raises there; here it simply clears the state)."""
        self.sequence_length = sequence_length
        if self._state is not None:
            self._reset_state(self._state.size(0), self._state.device)

    def _reset_state(self, B, device):
        if self._state is None or self._state.size(0) != B or self._state.device != device:
            self._state = torch.empty(B, 4, dtype=torch.int32, device=device)
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(self._lib.satrn_sift_reset(ptr(self._state), B, self.sos_id, stream), "satrn_sift_reset")
        self.batch_size = B

    @torch.no_grad()
    def sift(self, probs_step):
        three_d = probs_step.ndim != 2
        x = probs_step.squeeze(1) if three_d else probs_step
        if not x.is_cuda:
            raise SatrnError("DeviceDecodingManager.sift needs CUDA/HIP tensors (no CPU fallback)")
        x = x.float().contiguous()
        B, V = x.shape
        if V != self.vocab_size:
            raise SatrnError(f"vocabulary mismatch: {V} != {self.vocab_size}")
        if self._state is None or self._state.size(0) != B or self._state.device != x.device:
            self._reset_state(B, x.device)  # postprocessing.py:206-213: a changed batch size restarts the memories
        targets = torch.empty(B, dtype=torch.int64, device=x.device)
        probs = torch.empty(B, V, dtype=torch.float32, device=x.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(self._lib.satrn_sift(ptr(x), V, ptr(self._state), ptr(self.table(x.device)), B, V, ptr(targets), ptr(probs), V,
                                   stream), "satrn_sift")
        self._keep = x
        return targets, (probs.unsqueeze(1) if three_d else probs)


def decode(model, input, data_loader=None, expected=None, method="greedy", beam_width=3):
    """postprocessing/decoding.py:6-53: the inference / validation decoding switch.
    greedy -> ids int64 [B, L-1] (argmax of model(input, expected, False, 0.0), computed inside the decode kernel);
    beam -> model.beam_search(...) with max_sequence = expected.size(-1) - 1, int64 [B, max_sequence] on the CPU."""
    if method == "greedy":
        # the reference takes topk(1) of the returned step outputs; the decode kernel already produced that argmax
        # (logits, or the DecodingManager's masked probabilities -- the same winner either way)
        return model.greedy(input, expected.size(1) - 1)[1]
    if method == "beam":
        return model.beam_search(input=input, data_loader=data_loader, beam_width=beam_width,
                                 max_sequence=expected.size(-1) - 1)
    raise NotImplementedError(f"There's no '{method}' type yet.")
