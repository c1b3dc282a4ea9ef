"""Host-side mirror of the reference's module interface for the accelerated path.

Same constructor signatures, attribute tree (.encoder / .decoder / .criterion), forward signature and
state_dict keys as networks/EfficientSATRN.py:664-706 and networks/LiteSATRN.py:548-590, but every FLOP runs in
libsatrn_hip.so (hand-written gfx950 kernels) through the C-ABI of include/satrn_hip.h.  PyTorch only owns the
memory (flat parameter / gradient / buffer tensors, one workspace), the stream and autograd's outer graph.
"""
import ctypes
import math
import random

import torch
import torch.nn as nn

from . import _lib
from ._lib import SatrnError, check, ptr, satrn_config
from .utils import PAD, START

_DT = {"f32": 0, "fp32": 0, "float32": 0, "bf16": 1, "bfloat16": 1, 0: 0, 1: 1}


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# The stream the fused step / decode run on: high priority (the critical chain; the engine's weight-gradient stream is low priority),
# ONE per device and process.  A stream per MODEL made the pair (chain, engine side stream) of every later model land on hardware queues
# at the runtime's choice -- and pairs whose queues share a dispatch pipe do not overlap: the f32 leg of bench.py ran at 43 ms per step
# in the bench process against 23 ms alone, SwinTRN at 32 against 16 (tools/f32_in_process.py, tools/swin_in_process.py).
_CHAIN_STREAMS = {}


def _chain_stream(device):
    device = torch.device(device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    st = _CHAIN_STREAMS.get(key)
    if st is None:
        st = torch.cuda.Stream(device=device, priority=-1)
        _CHAIN_STREAMS[key] = st
    return st


class _Node(nn.Module):
    """Plain container node of the mirrored module tree (holds parameters / buffers / children only)."""


class SATRNCrossEntropy(nn.Module):
    """nn.CrossEntropyLoss(ignore_index=PAD) as the reference builds it (networks/EfficientSATRN.py:690-692) and
    calls it (train_modules/train_single_opt.py:82,86): input [B, V, T] (= logits.transpose(1, 2)), target [B, T]."""

    def __init__(self, ignore_index):
        super().__init__()
        self.ignore_index = int(ignore_index)

    def forward(self, input, target):
        return _CEFunction.apply(input, target, self.ignore_index)


class _CEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, target, pad_id):
        lib = _lib.load()
        if not input.is_cuda:
            raise SatrnError("SATRNCrossEntropy needs CUDA/HIP tensors (no CPU fallback)")
        lg = input.transpose(1, 2)
        if not lg.is_contiguous() or lg.dtype != torch.float32:
            lg = lg.contiguous().float()
        B, T, V = lg.shape
        if target.stride(1) != 1:
            target = target.contiguous()
        out = torch.zeros(4, dtype=torch.float32, device=lg.device)
        lse = torch.empty(B * T, dtype=torch.float32, device=lg.device)
        dl = torch.empty(B, T, V, dtype=torch.float32, device=lg.device)
        check(lib.satrn_cross_entropy(0, ptr(lg), ptr(target), target.stride(0), B, T, V, V, pad_id, ptr(out), ptr(lse),
                                      ptr(dl), _stream()), "satrn_cross_entropy")
        ctx.save_for_backward(dl)
        return out[2].clone()

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return (dl * g).transpose(1, 2), None, None


class _KDFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, outputs, labels, teacher_outputs, T, alpha):
        lib = _lib.load()
        if not outputs.is_cuda:
            raise SatrnError("loss_fn_kd needs CUDA/HIP tensors (no CPU fallback)")
        s = outputs.transpose(1, 2).contiguous().float()          # [B, T_len, V]
        t = teacher_outputs.transpose(1, 2).contiguous().float()
        B, TL, V = s.shape
        lab = labels if labels.stride(1) == 1 else labels.contiguous()
        out = torch.empty(1, dtype=torch.float32, device=s.device)
        dl = torch.empty_like(s)
        check(lib.satrn_kd_loss(ptr(s), ptr(t), ptr(lab), lab.stride(0), B, TL, V, float(T), float(alpha), ptr(out), ptr(dl),
                                _stream()), "satrn_kd_loss")
        ctx.save_for_backward(dl)
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return (dl * g).transpose(1, 2), None, None, None, None


def loss_fn_kd(outputs, labels, teacher_outputs, T=10, alpha=0.1):
    """train_modules/train_distillation.py:49-55 with the same call signature: outputs / teacher_outputs [B, V, T_len]
    (logits.transpose(1, 2)), labels [B, T_len]; one fused kernel computes the loss and d loss / d outputs."""
    return _KDFunction.apply(outputs, labels, teacher_outputs, T, alpha)


class _TFFunction(torch.autograd.Function):
    """Teacher-forced forward of the whole model as ONE autograd node; backward replays the engine's tape."""

    @staticmethod
    def forward(ctx, model, anchor, input, expected, record, teacher_forced):
        ctx.model = model
        ctx.gen = model._run_forward(input, expected, record=record, teacher_forced=teacher_forced)
        return model._last_logits

    @staticmethod
    def backward(ctx, dlogits):
        model = ctx.model
        if ctx.gen != model._gen:
            raise SatrnError("backward() of a stale forward: the engine keeps one tape (the latest forward)")
        model._run_backward(dlogits)
        return None, torch.zeros_like(model._anchor), None, None, None, None


class _SATRNBase(nn.Module):
    _NETWORK = 1
    _DEFAULT_DTYPE = "bf16"

    def _configure(self, c, FLAGS):
        """network-specific part of the configuration (the SATRN models read the encoder block of the YAML)"""
        e = FLAGS.SATRN.encoder
        c.enc_hidden, c.enc_filter, c.enc_heads, c.enc_layers = int(e.hidden_dim), int(e.filter_dim), int(e.head_num), int(e.layer_num)

    def __init__(self, FLAGS, train_dataset, checkpoint=None, decoding_manager=None, dtype=None):
        super().__init__()
        lib = _lib.load()
        self._lib = lib
        dt = _DT[self._DEFAULT_DTYPE if dtype is None else dtype]
        c = satrn_config()
        c.network = self._NETWORK
        c.rgb = int(FLAGS.data.rgb)
        c.height, c.width = int(FLAGS.input_size.height), int(FLAGS.input_size.width)
        d = FLAGS.SATRN.decoder
        c.dec_src, c.dec_hidden, c.dec_filter = int(d.src_dim), int(d.hidden_dim), int(d.filter_dim)
        c.dec_heads, c.dec_layers = int(d.head_num), int(d.layer_num)
        self._configure(c, FLAGS)
        c.num_classes = len(train_dataset.id_to_token)
        c.pad_id, c.sos_id = int(train_dataset.token_to_id[PAD]), int(train_dataset.token_to_id[START])
        c.dropout = float(FLAGS.dropout_rate)
        c.dtype = dt
        self._cfg = c
        self._h = lib.satrn_model_create(ctypes.byref(c))
        if not self._h:
            raise SatrnError("satrn_model_create: " + lib.satrn_last_error().decode())
        self.encoder = _Node()
        self.decoder = _Node()
        self._build_state()
        # attributes the reference's callers read
        self.decoder.layer_num = c.dec_layers
        self.decoder.st_id = c.sos_id
        self.decoder.pad_id = c.pad_id
        self.decoder.num_classes = c.num_classes
        self.decoder.hidden_dim = c.dec_hidden
        self.decoder.filter_dim = c.dec_filter
        # a reference DecodingManager (anything with .tokens / .rules) is compiled for the device once; like the
        # reference (SATRNDecoder.manager, networks/EfficientSATRN.py:464) it only acts in the inference decode loop
        from .decoding import DeviceDecodingManager
        self.decoder.manager = None if decoding_manager is None else DeviceDecodingManager.wrap(decoding_manager)
        if self.decoder.manager is not None and self.decoder.manager.vocab_size != c.num_classes:
            raise SatrnError("decoding_manager vocabulary does not match the dataset's")
        self.criterion = SATRNCrossEntropy(ignore_index=c.pad_id)
        self._anchor = torch.zeros(1, requires_grad=True)
        self._gen = 0
        self._ws = None
        self._ws_key = (0, 0)
        self._ws_cache = {}
        self._packed_version = -1
        self._bound = None
        self._last_logits = None
        self._stage = None
        self._dstage = None
        self._side = None
        self._warm = set()
        self._coin = None
        self.last_teacher_forced = True
        if checkpoint and checkpoint is not True:   # networks/SWIN.py:1025 has `checkpoint=True` as its default
            self.load_state_dict(checkpoint)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.satrn_model_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ------------------------------------------------------------------ state table -> module tree
    def _build_state(self):
        lib, h = self._lib, self._h
        n = lib.satrn_model_num_state(h)
        sizes = [lib.satrn_model_flat_size(h, k) for k in range(3)]
        flats = [torch.zeros(max(sizes[0], 1)), torch.zeros(max(sizes[1], 1)), torch.zeros(max(sizes[2], 1), dtype=torch.int64)]
        self._entries = []
        kind, ndim, init, fi, fo = (ctypes.c_int() for _ in range(5))
        shape = (ctypes.c_int64 * 4)()
        off = ctypes.c_int64()
        for i in range(n):
            name = lib.satrn_model_state_name(h, i).decode()
            check(lib.satrn_model_state_info(h, i, ctypes.byref(kind), ctypes.byref(ndim), shape, ctypes.byref(off),
                                             ctypes.byref(init), ctypes.byref(fi), ctypes.byref(fo)), "state_info")
            shp = tuple(int(shape[d]) for d in range(ndim.value))
            numel = 1
            for s in shp:
                numel *= s
            view = flats[kind.value][off.value: off.value + numel].view(shp)
            self._init_tensor(view, init.value, fi.value, fo.value)
            node = self
            parts = name.split(".")
            for p in parts[:-1]:
                if not hasattr(node, p):
                    node.add_module(p, _Node())
                node = getattr(node, p)
            if kind.value == 0:
                node.register_parameter(parts[-1], nn.Parameter(view))
            else:
                node.register_buffer(parts[-1], view)
            self._entries.append((name, kind.value, shp, off.value, numel, node, parts[-1]))
        self._flat = flats
        self._gflat = None

    @staticmethod
    def _init_tensor(t, init, fan_in, fan_out):
        with torch.no_grad():
            if init == 0:  # xavier_normal_
                t.normal_(0.0, math.sqrt(2.0 / float(fan_in + fan_out)))
            elif init in (1, 2):  # conv / linear default: kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in))
                b = 1.0 / math.sqrt(fan_in)
                t.uniform_(-b, b)
            elif init == 3:
                b = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
                t.uniform_(-b, b)
            elif init == 4:
                t.fill_(1)
            elif init == 5:
                t.zero_()
            elif init == 6:
                t.normal_(0.0, 1.0)
            elif init == 9:   # timm trunc_normal_(std=0.02): N(0, 0.02) cut at +-2 (a = -2, b = 2 are ~100 sigma: no-op bounds)
                t.normal_(0.0, 0.02).clamp_(-2.0, 2.0)
            elif init == 7:   # WindowAttention.relative_position_index (networks/SWIN.py:120-135)
                ws = int(round(math.sqrt(t.shape[0])))
                co = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
                rel = (co[:, :, None] - co[:, None, :]).permute(1, 2, 0).contiguous()
                rel[:, :, 0] += ws - 1
                rel[:, :, 1] += ws - 1
                rel[:, :, 0] *= 2 * ws - 1
                t.copy_(rel.sum(-1))
            elif init == 8:   # SwinTransformerBlock.attn_mask (networks/SWIN.py:288-309): 0 / -100 between the shifted regions
                nW, N = t.shape[0], t.shape[1]
                ws = int(round(math.sqrt(N)))
                res, shift = ws * int(round(math.sqrt(nW))), ws // 2
                img = torch.zeros(res, res)
                cnt = 0
                for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
                    for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
                        img[hs, wsl] = cnt
                        cnt += 1
                mw = img.view(res // ws, ws, res // ws, ws).permute(0, 2, 1, 3).reshape(-1, N)
                am = mw[:, None, :] - mw[:, :, None]
                t.copy_(torch.where(am != 0, torch.full_like(am, -100.0), torch.zeros_like(am)))

    def _tensor_of(self, entry):
        return getattr(entry[5], entry[6])

    # ------------------------------------------------------------------ device binding
    def _ensure_bound(self, device):
        if device.type != "cuda":
            raise SatrnError("the SATRN engine runs on MI355X only: move the model and inputs to 'cuda' (no CPU fallback)")
        first = self._tensor_of(self._entries[0])
        if self._bound == device and first.data_ptr() == self._flat[self._entries[0][1]].data_ptr() + 4 * self._entries[0][3]:
            return
        flats = [torch.zeros(f.numel(), dtype=f.dtype, device=device) for f in self._flat]
        with torch.no_grad():
            # while every tensor still is a view of the old flat buffers (the normal case) the upload is three copies, not two per
            # state entry (~1 600 small blit dispatches for EfficientSATRN, which a kernel trace of a short run then shows per "step")
            old = self._flat
            bulk = all(getattr(node, leaf).data_ptr() == old[kind].data_ptr() + off * old[kind].element_size() and getattr(node, leaf).dtype == old[kind].dtype
                       for name, kind, shp, off, numel, node, leaf in self._entries)
            if bulk:
                for k in range(3):
                    flats[k].copy_(old[k])
            for name, kind, shp, off, numel, node, leaf in self._entries:
                t = getattr(node, leaf)
                view = flats[kind][off: off + numel].view(shp)
                if not bulk:
                    view.copy_(t.detach().to(device=device, dtype=view.dtype))
                if kind == 0:
                    t.data = view
                    t.grad = None
                else:
                    node._buffers[leaf] = view
        self._flat = flats
        self._gflat = torch.zeros_like(flats[0])
        self._anchor = torch.zeros(1, device=device, requires_grad=True)
        check(self._lib.satrn_model_bind(self._h, ptr(flats[0]), ptr(self._gflat), ptr(flats[1]), ptr(flats[2])), "bind")
        # AdamW moments of the fused step: own flat tensors (NOT in the resizable workspace), moved with the model
        old = getattr(self, "_adam", None)
        self._adam = (torch.zeros_like(flats[0]), torch.zeros_like(flats[0]))
        if old is not None and old[0].numel() == flats[0].numel():
            self._adam[0].copy_(old[0])
            self._adam[1].copy_(old[1])
        check(self._lib.satrn_model_bind_optimizer(self._h, ptr(self._adam[0]), ptr(self._adam[1])), "bind_optimizer")
        self._bound = device
        self._packed_version = -1
        if self._ws is not None and self._ws.device != device:
            self._ws = None
            self._ws_key = (0, 0)

    def _ensure_ws(self, B, L, device):
        if self._ws is not None and B <= self._ws_key[0] and L <= self._ws_key[1]:
            return
        B2, L2 = max(B, self._ws_key[0]), max(L, self._ws_key[1])
        need = self._lib.satrn_model_workspace_bytes(self._h, B2, L2)
        # the old workspace stays allocated until set_workspace returns: the engine carries the dropout RNG word over from
        # it.  Nothing else that must survive lives there (Adam's moments / step count are bound separately), so a batch
        # that is longer than every earlier one only costs a re-pack of the compute weights
        old_ws = self._ws
        new_ws = torch.empty(need, dtype=torch.uint8, device=device)
        check(self._lib.satrn_model_set_workspace(self._h, ptr(new_ws), need, _stream()), "set_workspace")
        self._ws = new_ws
        del old_ws
        self._ws_key = (B2, L2)
        self._packed_version = -1
        self._stage = None
        self._dstage = None
        self._warm = set()

    def _param_version(self):
        ps = getattr(self, "_ptensors", None)
        if ps is None:  # the Parameter objects are created once (re-binding only swaps their .data)
            ps = self._ptensors = [self._tensor_of(e) for e in self._entries if e[1] == 0]
        v = 0
        for t in ps:
            v += t._version
        return v

    def _ensure_packed(self):
        v = self._param_version()
        if v != self._packed_version:
            check(self._lib.satrn_model_pack_weights(self._h, _stream()), "pack_weights")
            self._packed_version = v

    def _prepare(self, input, B, L):
        self._ensure_bound(input.device)
        self._ensure_ws(B, L, input.device)
        self._ensure_packed()

    def _img(self, input):
        c = self._cfg
        if input.dim() != 4 or tuple(input.shape[1:]) != (c.rgb, c.height, c.width):
            # the engine takes the geometry from the configuration: another shape would be read out of bounds
            raise SatrnError(f"input must be [B, {c.rgb}, {c.height}, {c.width}] (FLAGS.data.rgb / FLAGS.input_size), got {tuple(input.shape)}")
        if input.dtype != torch.float32 or not input.is_contiguous():
            input = input.float().contiguous()
        return input

    def reserve(self, B, L, device=None):
        """Size the workspace for batches up to B samples x L tokens now (data-parallel runs call this with the global
        maxima so that no rank re-allocates mid-run; growing later is safe, it only re-packs the compute weights)."""
        device = torch.device(device) if device is not None else (self._bound or next(self.parameters()).device)
        self._ensure_bound(device)
        self._ensure_ws(int(B), int(L), device)

    def optimizer_state_dict(self):
        """State of the fused AdamW (what torch.optim.AdamW.state_dict() holds per parameter, flat here) + the dropout RNG
        word; the reference checkpoints its optimizer the same way (train_modules/train_single_opt.py:497-512)."""
        if self._bound is None:
            raise SatrnError("optimizer_state_dict: the model has not run on a device yet")
        seed = ctypes.c_uint32(0)
        if self._ws is not None:
            check(self._lib.satrn_model_rng_state(self._h, ctypes.byref(seed), 0, _stream()), "rng_state")
        return {"exp_avg": self._adam[0].detach().clone(), "exp_avg_sq": self._adam[1].detach().clone(),
                "step": int(self._lib.satrn_model_get_step(self._h)), "rng": int(seed.value)}

    def load_optimizer_state_dict(self, sd):
        if self._bound is None:
            raise SatrnError("load_optimizer_state_dict: move the model to its device and run reserve() first")
        with torch.no_grad():
            self._adam[0].copy_(sd["exp_avg"])
            self._adam[1].copy_(sd["exp_avg_sq"])
        check(self._lib.satrn_model_set_step(self._h, int(sd["step"])), "set_step")
        if self._ws is not None and "rng" in sd:
            seed = ctypes.c_uint32(int(sd["rng"]))
            check(self._lib.satrn_model_rng_state(self._h, ctypes.byref(seed), 1, _stream()), "rng_state")

    def check_device_error(self):
        """raise if a kernel met token ids outside the embedding table / vocabulary since the last check (synchronises)"""
        bits = self._lib.satrn_device_error(_stream())
        if bits:
            raise SatrnError("token ids out of range reached the model ("
                             + ("decoder input outside the embedding table; " if bits & 1 else "")
                             + ("loss target outside the vocabulary; " if bits & 2 else "")
                             + "rewrite the loader's -1 padding to <PAD> before the forward, as train_single_opt.py:78 does)")

    # ------------------------------------------------------------------ engine calls
    def _run_forward(self, input, expected, record, teacher_forced=True):
        input = self._img(input)
        expected = expected.contiguous()
        B, L = expected.shape
        self._prepare(input, B, L)
        logits = torch.empty(B, L - 1, self._cfg.num_classes, dtype=torch.float32, device=input.device)
        self._gen += 1
        check(self._lib.satrn_model_forward(self._h, ptr(input), ptr(expected), B, L, int(self.training), int(record),
                                            int(teacher_forced), ptr(logits), _stream()), "satrn_model_forward")
        self._last_logits = logits
        self._keep = (input, expected)  # the tape reads them in backward
        return self._gen

    def _attach_grads(self):
        """param.grad views into the flat gradient buffer; a None grad (after zero_grad) means 'start from zero'."""
        fresh = False
        for name, kind, shp, off, numel, node, leaf in self._entries:
            if kind != 0:
                continue
            p = getattr(node, leaf)
            if p.grad is None or p.grad.data_ptr() != self._gflat.data_ptr() + 4 * off:
                fresh = fresh or p.grad is None
                p.grad = self._gflat[off: off + numel].view(shp)
        return fresh

    def _run_backward(self, dlogits):
        first = self._tensor_of(self._entries[0])
        if first.grad is None:
            self._gflat.zero_()
        self._attach_grads()
        dl = dlogits.contiguous().float()
        check(self._lib.satrn_model_backward(self._h, ptr(dl), _stream()), "satrn_model_backward")

    def forward(self, input, expected, is_train, teacher_forcing_ratio):
        """networks/EfficientSATRN.py:697-706: -> [B, L-1, V] (teacher-forced logits, or greedy-step logits)."""
        if is_train:
            # the reference's coin (Python `random`, :489): teacher forced, or autoregressive WITH gradients (:496-525)
            # (data-parallel ranks share the coin: set_coin(dp.SharedCoin(seed)) -- every rank must take the same branch in a step)
            tf = (self._coin if self._coin is not None else random).random() < teacher_forcing_ratio
            self._ensure_bound(input.device)
            return _TFFunction.apply(self, self._anchor, input, expected, torch.is_grad_enabled(), tf)
        return self.greedy(input, expected.size(1) - 1)[0]

    @torch.no_grad()
    def greedy(self, input, num_steps, use_graph=False, forced=None):
        """networks/EfficientSATRN.py:528-561: -> (logits [B, steps, V], ids [B, steps]); with a decoding manager the
        first tensor holds the masked softmax probabilities instead (:553-554), the rules run inside the decode kernel.
        With use_graph the whole decode (encoder + every step) replays as one hipGraph from persistent staging buffers
        (measured SLOWER than eager launches on ROCm 7.2 for this ~10^4-node graph: 100 ms vs 80 ms per 64x231 batch, so
        it is off by default); the first call of a shape always runs eagerly.
        forced (int64 [B, steps]): forced replay -- step t + 1 is fed forced[:, t] instead of its own argmax (the returned ids
        stay the argmax); with a reference decode's ids every step's logits can be compared, not only a prefix."""
        input = self._img(input)
        B = input.size(0)
        self._prepare(input, B, num_steps + 1)
        V = self._cfg.num_classes
        key = (B, num_steps, tuple(input.shape))
        if self._dstage is None or self._dstage[0] != key:
            self._dstage = (key, torch.empty_like(input), torch.empty(B, num_steps, V, dtype=torch.float32, device=input.device),
                            torch.empty(B, num_steps, dtype=torch.int64, device=input.device), [False])
        _, simg, slog, sids, warm = self._dstage
        simg.copy_(input)
        cur = torch.cuda.current_stream()
        if self._side is None:
            self._side = _chain_stream(input.device)  # high priority: the critical chain; shared by the models of the process
        self._side.wait_stream(cur)
        mgr = self.decoder.manager
        with torch.cuda.stream(self._side):
            if forced is not None:
                if mgr is not None:
                    raise SatrnError("forced replay runs without a decoding manager")
                fids = forced.to(device=input.device, dtype=torch.int64).contiguous()
                if tuple(fids.shape) != (B, num_steps):
                    raise SatrnError(f"forced ids must be [B, steps] = [{B}, {num_steps}], got {tuple(fids.shape)}")
                self._keep_forced = fids
                check(self._lib.satrn_model_greedy_forced(self._h, ptr(simg), None, B, num_steps, ptr(fids), ptr(slog), ptr(sids),
                                                          _stream()), "satrn_model_greedy_forced")
            elif mgr is not None:
                check(self._lib.satrn_model_greedy_rules(self._h, ptr(simg), None, B, num_steps, ptr(mgr.table(input.device)),
                                                         ptr(slog), ptr(sids), _stream()), "satrn_model_greedy_rules")
            else:
                check(self._lib.satrn_model_greedy(self._h, ptr(simg), None, B, num_steps, ptr(slog), ptr(sids),
                                                   int(use_graph and warm[0]), _stream()), "satrn_model_greedy")
        cur.wait_stream(self._side)
        warm[0] = True
        return slog.clone(), sids.clone()

    # ------------------------------------------------------------------ diagnostics: stage-boundary probes
    def enable_probes(self, on=True):
        """the next forwards remember the activation at every stage boundary (read with probes()) and, for buffers registered by
        probe_grads() before the backward, the gradient wrt it"""
        check(self._lib.satrn_model_probe_enable(self._h, int(on)), "probe_enable")

    def _probe_info(self):
        out = []
        name, rows, cols = ctypes.c_char_p(), ctypes.c_int64(), ctypes.c_int()
        for i in range(self._lib.satrn_model_probe_count(self._h)):
            check(self._lib.satrn_model_probe_info(self._h, i, ctypes.byref(name), ctypes.byref(rows), ctypes.byref(cols)), "probe_info")
            out.append((name.value.decode(), rows.value, cols.value))
        return out

    def probes(self):
        """-> {name: fp32 tensor [rows, cols]}: the stage-boundary activations of the last forward"""
        res = {}
        for i, (name, rows, cols) in enumerate(self._probe_info()):
            t = torch.empty(rows, cols, dtype=torch.float32, device=self._bound)
            check(self._lib.satrn_model_probe_read(self._h, i, ptr(t), _stream()), "probe_read")
            res[name] = t
        return res

    def probe_grads(self):
        """call between forward and backward: -> {name: fp32 tensor} that the backward fills with d loss / d activation"""
        res = {}
        for i, (name, rows, cols) in enumerate(self._probe_info()):
            t = torch.zeros(rows, cols, dtype=torch.float32, device=self._bound)
            check(self._lib.satrn_model_probe_set_grad(self._h, i, ptr(t)), "probe_set_grad")
            res[name] = t
        self._probe_keep = res
        return res

    def set_coin(self, coin):
        """source of the per-batch teacher-forcing coin (anything with .random() -> [0, 1)); None = Python's global `random`,
        as the reference (networks/EfficientSATRN.py:489).  Data-parallel ranks pass dp.SharedCoin(seed)."""
        self._coin = coin

    def last_decode_path(self):
        """-> (path, giveups, note): which kernel produced the last greedy result -- "pipe" (one persistent workgroup per decoder
        role), "per_image" (one workgroup per image) or "stepwise"; giveups = pipelines that timed out and were re-run; note =
        why the pipeline was not taken."""
        g = ctypes.c_int(0)
        path = self._lib.satrn_model_last_decode_path(self._h, ctypes.byref(g))
        note = self._lib.satrn_model_decode_note(self._h)
        return {0: "none", 1: "pipe", 2: "per_image", 3: "stepwise"}.get(path, "?"), g.value, (note.decode() if note else "")

    def _feat_tokens(self):
        c = self._cfg
        if self._NETWORK == 2:
            return ((c.height // c.swin_patch) >> 3) ** 2
        f = 32 if self._NETWORK == 1 else 16
        return (c.height // f) * (c.width // f)

    @torch.no_grad()
    def encode(self, input):
        """SATRNEncoder.forward (networks/EfficientSATRN.py:311-323) in eval mode -> [B, hw, c] fp32."""
        input = self._img(input)
        B = input.size(0)
        self._prepare(input, B, 2)
        n = self._feat_tokens()
        src = torch.empty(B, n, self._cfg.enc_hidden, dtype=torch.float32, device=input.device)
        check(self._lib.satrn_model_encode(self._h, ptr(input), B, ptr(src), _stream()), "satrn_model_encode")
        return src

    # ------------------------------------------------------------------ fused training step (bench / trainer fast path)
    def train_step(self, input, expected, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-6, max_grad_norm=2.0,
                   grad_scale=1.0, use_graph=False, phase=3, bn_eval=False, teacher_forcing_ratio=1.0, teacher_forced=None):
        """forward + CE + backward + clip_grad_norm_ + AdamW + weight re-pack in ONE library call
        (train_modules/train_single_opt.py:80-98 with teacher forcing).  Default: eager launches on two HIP streams (weight
        gradients run beside the data-gradient chain; measured 10.4 ms vs 13.1 ms for the single-chain hipGraph replay,
        which use_graph=True selects; the autoregressive branch replays in 112 ms vs 88 ms eager).  phase: 1 = forward/backward only,
        2 = clip + AdamW only (data-parallel callers all-reduce the flat gradient in between), 3 = both; 16 + k = backward
        segment k of phase 1 (k = 0..3 in order; overlapped gradient exchange, see dp.dp_train_step).
        lr = (encoder_lr, decoder_lr) and/or weight_decay = (encoder_wd, decoder_wd) select the reference's DUAL-optimizer
        iteration (train_modules/train_dual_opt.py:87-113): encoder.* and decoder.* gradients are clipped separately and
        stepped with their own learning rates (eager only; plain Adam = weight_decay 0, what that trainer uses).
        bn_eval=True: module.eval() semantics with gradients (BatchNorm running statistics, no dropout; eager) -- every
        sample independent of its batch, the mode of the data-parallel equivalence test.
        teacher_forcing_ratio < 1: the reference's per-batch coin (networks/EfficientSATRN.py:489; Python's `random`, or the coin given
        to set_coin -- data-parallel ranks share one, dp.SharedCoin) decides between the teacher-forced decoder and the
        autoregressive one WITH gradients (:496-525); the coin is flipped by the call that starts a step (phase bit 0 / segment 0).
        teacher_forced=True/False overrides the coin (the later calls of a split step must repeat the first call's branch:
        last_teacher_forced)."""
        starts = (int(phase) & 31) in (1, 3) or (int(phase) & 16 and int(phase) & 3 == 0)
        if teacher_forced is None:
            if starts:
                # one draw per training batch whatever the ratio, as the reference (networks/EfficientSATRN.py:488-489) and forward():
                # the Python RNG stream -- and a SharedCoin's flip count -- stay in step with them when a run mixes ratio 1.0 with < 1
                teacher_forced = (self._coin if self._coin is not None else random).random() < teacher_forcing_ratio
            else:
                teacher_forced = self.last_teacher_forced
        self.last_teacher_forced = bool(teacher_forced)
        if not teacher_forced:
            phase = int(phase) | 64
        if bn_eval:
            phase = int(phase) | 32
            use_graph = False
        dual = isinstance(lr, (tuple, list)) or isinstance(weight_decay, (tuple, list))
        if dual:
            lr = tuple(lr) if isinstance(lr, (tuple, list)) else (lr, lr)
            weight_decay = tuple(weight_decay) if isinstance(weight_decay, (tuple, list)) else (weight_decay, weight_decay)
            if use_graph:
                raise ValueError("the dual-optimizer step runs eagerly (use_graph=False)")
        input = self._img(input)
        B, L = expected.shape
        self._ensure_bound(input.device)
        self._ensure_ws(B, L, input.device)
        # the in-step AdamW re-packs the compute copies itself and does not bump torch's tensor versions: a version that
        # moved since the last step means load_state_dict / an EMA / a manual edit touched the masters -> pack again
        if self._packed_version != self._param_version():
            self._ensure_packed()
        if self._stage is None or self._stage[0].shape != input.shape or self._stage[1].shape != expected.shape:
            self._stage = (torch.empty_like(input), torch.empty_like(expected.contiguous()))
        if starts:  # calls that start a step stage its inputs
            self._stage[0].copy_(input, non_blocking=True)
            self._stage[1].copy_(expected, non_blocking=True)
        if dual:
            hy = (ctypes.c_float * 9)(lr[0], betas[0], betas[1], eps, weight_decay[0], max_grad_norm, 0.0, 0.0, grad_scale)
            hy2 = (ctypes.c_float * 9)(lr[1], betas[0], betas[1], eps, weight_decay[1], max_grad_norm, 0.0, 0.0, grad_scale)
        else:
            hy = (ctypes.c_float * 9)(lr, betas[0], betas[1], eps, weight_decay, max_grad_norm, 0.0, 0.0, grad_scale)
        # graphs are captured on a private stream (the legacy default stream cannot capture); the first step of a shape
        # runs eagerly so that one-time kernel attribute setup never lands inside a capture
        key = (B, L, int(phase))
        warm = key in self._warm
        self._warm.add(key)
        cur = torch.cuda.current_stream()
        if self._side is None:
            self._side = _chain_stream(input.device)  # high priority: the critical chain; shared by the models of the process
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            if dual:
                check(self._lib.satrn_model_train_step_dual(self._h, ptr(self._stage[0]), ptr(self._stage[1]), B, L, hy, hy2,
                                                            int(phase), _stream()), "satrn_model_train_step_dual")
            else:
                check(self._lib.satrn_model_train_step(self._h, ptr(self._stage[0]), ptr(self._stage[1]), B, L, hy,
                                                       int(use_graph and warm), int(phase), _stream()), "satrn_model_train_step")
        cur.wait_stream(self._side)
        self._gen += 1
        self._packed_version = self._param_version()  # parameters were updated and re-packed inside the step

    def profile_step(self, input, expected):
        """One eager forward + CE + backward with HIP events around every launch -> list of per-kernel-family dicts
        (kernel, launches, ms, flops, bytes), sorted by time."""
        import json
        input = self._img(input)
        expected = expected.contiguous()
        B, L = expected.shape
        self._prepare(input, B, L)
        buf = ctypes.create_string_buffer(1 << 18)
        check(self._lib.satrn_model_profile_step(self._h, ptr(input), ptr(expected), B, L, buf, len(buf), _stream()),
              "satrn_model_profile_step")
        self._gen += 1
        return json.loads(buf.value.decode())

    def segment_range(self, seg):
        """[lo, hi) of the flat gradient that train_step(phase=16 + seg) completes (see include/satrn_hip.h)."""
        lo, hi = ctypes.c_int64(), ctypes.c_int64()
        check(self._lib.satrn_model_segment_range(self._h, int(seg), ctypes.byref(lo), ctypes.byref(hi)), "segment_range")
        return lo.value, hi.value

    def flat_grad(self):
        """the flat fp32 gradient buffer every p.grad is a view of (what the data-parallel all-reduce runs on)."""
        if self._gflat is None:
            self._ensure_bound(next(self.parameters()).device)
        return self._gflat

    def flat_params(self):
        if self._gflat is None:
            self._ensure_bound(next(self.parameters()).device)
        return self._flat[0]

    def read_loss(self):
        """-> (mean loss, valid-token count, grad-norm) of the last train_step / loss pass (synchronises)."""
        out = (ctypes.c_float * 4)()
        check(self._lib.satrn_model_read_loss(self._h, out, _stream()), "read_loss")  # fails on flagged out-of-range token ids
        return float(out[2]), float(out[1]), math.sqrt(max(float(out[3]), 0.0))

    def last_sequence(self, B, L):
        """-> int64 [B, L-1]: argmax of the last forward's / train_step's logits (the `sequence` of
        train_modules/train_single_opt.py:82-84, what StepMetrics.update takes).  Valid until the model runs again."""
        ids = torch.empty(B, L - 1, dtype=torch.int64, device=self._bound)
        check(self._lib.satrn_model_last_sequence(self._h, ptr(ids), int(B), int(L), _stream()), "satrn_model_last_sequence")
        return ids

    def read_grad_norms(self):
        """-> (encoder grad-norm, decoder grad-norm) of the last dual-optimizer train_step (train_dual_opt.py:101-109)."""
        out = (ctypes.c_float * 2)()
        check(self._lib.satrn_model_read_grad_norms(self._h, out, _stream()), "read_grad_norms")
        return math.sqrt(max(float(out[0]), 0.0)), math.sqrt(max(float(out[1]), 0.0))


class EfficientSATRN(_SATRNBase):
    """Drop-in for networks/EfficientSATRN.py:664 (EfficientNetV2-S backbone, /32)."""
    _NETWORK = 1

    @torch.no_grad()
    def beam_search(self, input, data_loader, topk=1, beam_width=5, max_sequence=230):
        """networks/EfficientSATRN.py:708-867 -> int64 [B, max_sequence] on the CPU (what id_to_string consumes): per image
        a best-first search (priority queue on -sum(log p)/len) that ends at the first popped <EOS> or after
        max_sequence-1 expansions; rows start with <SOS> and are padded with <PAD>, as the reference returns them.  The whole
        batch runs in one launch, one workgroup per image (satrn_model_beam_search).  Only topk=1: the reference's own
        packing of the result (:857-865) cannot represent more than one utterance per image."""
        if topk != 1:
            raise NotImplementedError("beam_search: topk must be 1 (the reference's output packing accepts nothing else)")
        t2i = data_loader.dataset.token_to_id
        eos, pad = int(t2i["<EOS>"]), int(t2i["<PAD>"])
        if int(t2i["<SOS>"]) != self._cfg.sos_id:
            raise ValueError("data_loader's <SOS> id differs from the model's")
        input = self._img(input)
        B = input.size(0)
        self._prepare(input, B, 2)
        seq = torch.empty(B, int(max_sequence), dtype=torch.int64, device=input.device)
        check(self._lib.satrn_model_beam_search(self._h, ptr(input), B, int(beam_width), int(max_sequence), eos, pad, ptr(seq),
                                                _stream()), "satrn_model_beam_search")
        return seq.cpu()


class LiteSATRN(_SATRNBase):
    """Drop-in for networks/LiteSATRN.py:548 (ShallowCNN backbone, /16)."""
    _NETWORK = 0


class SWIN(_SATRNBase):
    """Drop-in for networks/SWIN.py:1024 (BASELINE configs[3], SwinTRN): Swin-B/384 encoder (patch 4, embed 128, depths
    2/2/18/2, heads 4/8/16/32, window 12, shifted windows with the 0 / -100 mask, relative position bias, absolute position
    embedding, stochastic depth up to 0.5, exact-erf GELU MLP, patch merging, final LayerNorm -> [B, 144, 1024]) + the
    transformer decoder of the SATRN models with the YAML's decoder dims (configs/SWIN.yaml:11-16).  The reference hard-codes
    the encoder geometry and downloads ImageNet-22k weights at construction (:1028-1034; no network here: random init, load a
    checkpoint instead); `swin=dict(embed_dim=, depths=, num_heads=, window_size=, patch_size=, drop_path_rate=, head_classes=)`
    overrides the geometry for tests.  state_dict keys == the reference module's (encoder.* incl. the unused head and the
    relative_position_index / attn_mask buffers, decoder.*)."""
    _NETWORK = 2
    _SWIN_DEFAULT = dict(embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32), window_size=12, patch_size=4, drop_path_rate=0.5,
                         head_classes=21841)

    def __init__(self, FLAGS, train_dataset, checkpoint=True, decoding_manager=None, dtype=None, swin=None):
        self._swin = dict(self._SWIN_DEFAULT, **(swin or {}))
        super().__init__(FLAGS, train_dataset, checkpoint, decoding_manager, dtype)

    def _configure(self, c, FLAGS):
        g = self._swin
        c.swin_embed, c.swin_window, c.swin_patch = int(g["embed_dim"]), int(g["window_size"]), int(g["patch_size"])
        c.swin_head_classes, c.swin_drop_path = int(g["head_classes"]), float(g["drop_path_rate"])
        for i in range(4):
            c.swin_depths[i], c.swin_heads[i] = int(g["depths"][i]), int(g["num_heads"][i])
        # the encoder block of SWIN.yaml is unused by the reference (networks/SWIN.py:1028-1031); the engine's generic checks
        # want a consistent width
        c.enc_hidden, c.enc_filter, c.enc_heads, c.enc_layers = c.dec_src, c.dec_src, 1, 0


class _Half(nn.Module):
    """Shared plumbing of the encoder-only / decoder-only wrappers: the engine always holds the whole model (one flat
    parameter buffer); the wrapper registers only its half in the module tree, so state_dict() has exactly the
    reference's keys ('encoder.*' or 'decoder.*') and the other half keeps its initial values, unused."""

    def __init__(self, FLAGS, train_dataset, dtype):
        super().__init__()
        # not a registered submodule: its parameters must not appear in this wrapper's state_dict
        object.__setattr__(self, "_full", EfficientSATRN(FLAGS, train_dataset, None, None, dtype))

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        self._full._apply(fn, *a, **k)  # .to(device) / .float() reach the unregistered half as well
        return self


class EfficientSATRN_encoder(_Half):
    """networks/EfficientSATRN.py:870-894: encoder-only wrapper used by the ensemble path; forward(input) -> [B, N, D]."""

    def __init__(self, FLAGS, train_dataset, checkpoint=None, dtype=None):
        super().__init__(FLAGS, train_dataset, dtype)
        self.encoder = self._full.encoder
        if checkpoint:
            self.load_state_dict(checkpoint)

    def forward(self, input):
        return self._full.encode(input)


class EfficientSATRN_decoder(_Half):
    """networks/EfficientSATRN.py:897-952: decoder-only wrapper with the step-wise ensemble interface.

    step_forward(src, target) -> [B, 1, V] consumes one token per sequence and keeps the per-layer history on the device
    (self-attention K/V cache; the cross-attention K/V of `src` are projected once, at the first step after
    reset_status() -- the reference re-projects them every step, :386-396).  `src` must be the same encoder output for
    every step of a sequence, as in utils/ensemble_utils.py:84-96.  max_steps bounds the history (default 256; the
    reference's drivers use max_sequence + 1 = 231)."""

    def __init__(self, FLAGS, train_dataset, checkpoint=None, dtype=None, max_steps=256):
        super().__init__(FLAGS, train_dataset, dtype)
        self.decoder = self._full.decoder
        self.step_idx = 0
        self.max_steps = int(max_steps)
        if checkpoint:
            self.load_state_dict(checkpoint)

    @torch.no_grad()
    def step_forward(self, src, target):
        full = self._full
        if not src.is_cuda:
            raise SatrnError("the SATRN engine runs on MI355X only: move src/target to 'cuda' (no CPU fallback)")
        B = src.size(0)
        cur = torch.cuda.current_stream()
        if self.step_idx == 0:
            src = src.float().contiguous()
            dummy = torch.empty(B, 1, 1, 1, device=src.device)
            full._prepare(dummy, B, self.max_steps + 1)
            check(full._lib.satrn_model_step_begin(full._h, ptr(src), B, self.max_steps, _stream()), "satrn_model_step_begin")
            self._keep_src = src
        target = target.reshape(B).to(torch.int64).contiguous()
        logits = torch.empty(B, 1, full._cfg.num_classes, dtype=torch.float32, device=src.device)
        check(full._lib.satrn_model_step(full._h, ptr(target), ptr(logits), _stream()), "satrn_model_step")
        self._keep_tgt = target  # the launch reads it asynchronously
        self.step_idx += 1
        return logits

    def reset_status(self):
        self.step_idx = 0


