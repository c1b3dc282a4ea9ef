"""The library's run-time switches live in FOUR environment variables (csrc/kernels.h): SATRN_OFF (features whose current form is
switched off: the form it replaced runs), SATRN_KNOBS (name=value tuning knobs, opt-in forms, tri-state routes), SATRN_PROF (diagnostics),
SATRN_TIMING (timing experiments that skip work).  These helpers edit the lists in os.environ; the library reads them per call."""
import os


def _tokens(var):
    return [t for t in os.environ.get(var, "").replace(" ", ",").split(",") if t]


def _store(var, toks):
    if toks:
        os.environ[var] = ",".join(toks)
    else:
        os.environ.pop(var, None)


def off(*names):
    """switch the named features off (the forms they replaced run)"""
    t = _tokens("SATRN_OFF")
    _store("SATRN_OFF", t + [n for n in names if n not in t])


def on(*names):
    """... and back on"""
    _store("SATRN_OFF", [t for t in _tokens("SATRN_OFF") if t not in names])


def is_off(name):
    return name in _tokens("SATRN_OFF")


def knob(name, value=None):
    """set (value) or remove (None) a knob"""
    t = [x for x in _tokens("SATRN_KNOBS") if x.split("=")[0] != name]
    if value is not None:
        t.append(f"{name}={value}")
    _store("SATRN_KNOBS", t)


def prof(*names):
    _store("SATRN_PROF", list(names))


def snapshot():
    return {v: os.environ.get(v) for v in ("SATRN_OFF", "SATRN_KNOBS", "SATRN_PROF", "SATRN_TIMING")}


def restore(snap):
    for v, val in snap.items():
        if val is None:
            os.environ.pop(v, None)
        else:
            os.environ[v] = val
