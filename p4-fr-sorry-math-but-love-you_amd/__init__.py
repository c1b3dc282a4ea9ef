"""MI355X-native SATRN hot path: HIP kernels + C-ABI (csrc/, libsatrn_hip.so) and the host-side mirror of the
reference's module interface (networks.py, utils.py).  No CPU fallback: every compute path goes through the library."""
from . import _lib
from ._lib import SatrnError
from .networks import EfficientSATRN, LiteSATRN, SWIN, EfficientSATRN_encoder, EfficientSATRN_decoder, SATRNCrossEntropy, loss_fn_kd
from . import decoding
from . import switches
from . import metrics
from .metrics import StepMetrics
from .decoding import DeviceDecodingManager, compile_rules, decode
from .preprocess import preprocess_images
from .utils import get_network, load_vocab, Flags, id_to_string, START, END, PAD, SPECIAL_TOKENS

__all__ = ["EfficientSATRN", "LiteSATRN", "SWIN", "EfficientSATRN_encoder", "EfficientSATRN_decoder", "SATRNCrossEntropy",
           "get_network", "load_vocab", "Flags", "id_to_string", "decode", "StepMetrics", "DeviceDecodingManager", "loss_fn_kd", "preprocess_images", "SatrnError", "START", "END", "PAD", "SPECIAL_TOKENS"]
