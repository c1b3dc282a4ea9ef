"""Per-step training metrics on the device.

The reference's loop turns every batch into Python strings (utils/utils.py:134-164, one .item() per token) to compute the
word error rate, the sentence accuracy and the symbol accuracy (train_modules/train_single_opt.py:101-109,
utils/metrics.py:9-34).  Here the same numbers are accumulated by one kernel launch per batch over the token ids; nothing
is read back until result() is called."""
import ctypes

import torch

from . import _lib
from ._lib import SatrnError, check, ptr
from .utils import END, PAD, START


class StepMetrics:
    def __init__(self, token_to_id):
        self._ids = (int(token_to_id[PAD]), int(token_to_id[START]), int(token_to_id[END]), int(token_to_id.get("", -1)))
        self._acc = None
        self._lib = _lib.load()

    def reset(self):
        if self._acc is not None:
            self._acc.zero_()

    @torch.no_grad()
    def update(self, sequence, expected):
        """sequence [B, T] predicted ids (e.g. logits.argmax(-1)), expected [B, L] = <SOS> tokens <EOS> <PAD>/-1 ...;
        asynchronous (no host synchronisation)."""
        if not sequence.is_cuda or not expected.is_cuda:
            raise SatrnError("StepMetrics needs CUDA/HIP tensors (no CPU fallback)")
        seq = sequence.to(torch.int64)
        exp = expected.to(torch.int64)
        if seq.stride(-1) != 1:
            seq = seq.contiguous()
        if exp.stride(-1) != 1:
            exp = exp.contiguous()
        if self._acc is None or self._acc.device != seq.device:
            self._acc = torch.zeros(5, dtype=torch.float64, device=seq.device)
        B, T = seq.shape
        L = exp.size(1)
        pad, sos, eos, empty = self._ids
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(self._lib.satrn_step_metrics(ptr(seq), seq.stride(0), T, ptr(exp), exp.stride(0), L, B, pad, sos, eos, empty,
                                           ptr(self._acc), stream), "satrn_step_metrics")
        self._keep = (seq, exp)

    def result(self):
        """-> dict(wer, sentence_acc, symbol_acc, sentences, correct_symbols, total_symbols); synchronises."""
        if self._acc is None:
            return dict(wer=0.0, sentence_acc=0.0, symbol_acc=0.0, sentences=0, correct_symbols=0, total_symbols=0)
        s_wer, n, ok, cs, ts = self._acc.tolist()
        n1 = max(n, 1.0)
        return dict(wer=s_wer / n1, sentence_acc=ok / n1, symbol_acc=cs / max(ts, 1.0), sentences=int(n),
                    correct_symbols=int(cs), total_symbols=int(ts), sum_wer=s_wer, correct_sentences=int(ok))
