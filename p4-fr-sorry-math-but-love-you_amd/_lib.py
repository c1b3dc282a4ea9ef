"""ctypes binding of libsatrn_hip.so.  Signatures are read from include/satrn_hip.h (the C-ABI contract), so the
binding cannot drift from the header.  There is NO fallback: without the HIP library every product path raises."""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(HERE, "..", "include", "satrn_hip.h")
LIBPATH = os.path.join(HERE, "libsatrn_hip.so")

_SCALARS = {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "size_t": ctypes.c_size_t,
            "uint32_t": ctypes.c_uint32, "int64_t": ctypes.c_int64, "void": None}


class SatrnError(RuntimeError):
    pass


class satrn_config(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in
                ("network", "rgb", "height", "width", "enc_hidden", "enc_filter", "enc_heads", "enc_layers", "dec_src",
                 "dec_hidden", "dec_filter", "dec_heads", "dec_layers", "num_classes", "pad_id", "sos_id")] + \
               [("dropout", ctypes.c_float), ("dtype", ctypes.c_int), ("swin_embed", ctypes.c_int), ("swin_depths", ctypes.c_int * 4),
                ("swin_heads", ctypes.c_int * 4), ("swin_window", ctypes.c_int), ("swin_patch", ctypes.c_int),
                ("swin_head_classes", ctypes.c_int), ("swin_drop_path", ctypes.c_float)]


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every function the header declares."""
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    txt = re.sub(r"typedef struct satrn_config \{.*?\} satrn_config;", " ", txt, flags=re.S)
    out = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ ]*?[\s\*]+)(satrn_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()

        def ctype(t):
            t = t.strip()
            if "*" in t:
                return ctypes.c_char_p if t.replace(" ", "") == "constchar*" else ctypes.c_void_p
            base = t.replace("const", "").strip().split()[0]
            return _SCALARS[base]
        argt = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                # drop the parameter name (last identifier) unless the token is a bare type
                mm = re.match(r"(.*?)([A-Za-z_][A-Za-z0-9_]*)?$", a)
                typ = a if "*" in a and a.rstrip().endswith("*") else re.sub(r"\s*[A-Za-z_][A-Za-z0-9_]*$", "", a)
                argt.append(ctype(typ))
        out[name] = (ctype(ret), argt)
    return out


_lib = None
_sigs = None


def load():
    """Load the library (once).  Raises SatrnError if it has not been built."""
    global _lib, _sigs
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBPATH):
        raise SatrnError(f"{LIBPATH} is missing: build it with `python __graft_entry__.py` (hipcc, gfx950). "
                         "There is no CPU fallback for the SATRN hot path.")
    lib = ctypes.CDLL(LIBPATH)
    _sigs = parse_header()
    for name, (ret, args) in _sigs.items():
        fn = getattr(lib, name)  # AttributeError if the header declares a symbol the library lacks
        fn.restype = ret
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().satrn_last_error()
        raise SatrnError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """device pointer of a torch tensor (or None)"""
    return None if t is None else ctypes.c_void_p(t.data_ptr())
