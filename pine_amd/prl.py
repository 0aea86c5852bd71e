"""ctypes binding of the PRL front-end (include/pine_prl.h, libpine_prl.so): run .pine scripts."""
import ctypes as C
import os

import numpy as np

from . import _lib  # loads libpine_gpu.so first (libpine_prl.so links against it)

LIB_PATH = os.environ.get("PINE_PRL_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libpine_prl.so"))
DRY_RUN, ECHO = 1, 2

SIGNATURES = {
    "pine_prl_interpret": (C.c_int, [C.c_char_p, C.c_int, C.c_int]),
    "pine_prl_output": (C.c_char_p, []),
    "pine_prl_last_error": (C.c_char_p, []),
    "pine_prl_last_film": (C.c_int64, [C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pine_prl_eval": (C.c_int64, [C.c_char_p, C.c_char_p, C.c_int64]),
}


class PrlError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    C.CDLL(_lib.LIB_PATH, mode=C.RTLD_GLOBAL)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()


def interpret(source, dry_run=False, echo=False, device=0):
    """Run a whole script; returns what it printed (plus the @render / @save records of a dry run)."""
    rc = lib.pine_prl_interpret(source.encode(), (DRY_RUN if dry_run else 0) | (ECHO if echo else 0), int(device))
    out = lib.pine_prl_output().decode()
    if rc < 0:
        raise PrlError(lib.pine_prl_last_error().decode())
    return out


def last_film():
    """Float film (H, W, 4) of the last render() of the last interpret() call on this thread."""
    p, w, h = C.POINTER(C.c_float)(), C.c_int(0), C.c_int(0)
    n = lib.pine_prl_last_film(C.byref(p), C.byref(w), C.byref(h))
    if n == 0:
        return None
    return np.ctypeslib.as_array(p, shape=(h.value, w.value, 4)).copy()


def evaluate(expression):
    """'<type> <value>' of one expression (floats as hex floats): a test hook."""
    buf = C.create_string_buffer(1 << 14)
    n = lib.pine_prl_eval(expression.encode(), buf, len(buf))
    if n < 0:
        raise PrlError(lib.pine_prl_last_error().decode())
    return buf.value.decode()


def scene_of_dry_run(output):
    """The .pscene text and (spp, max_path_length) of the first @render record of a dry-run output."""
    lines = output.splitlines()
    i = next(k for k, l in enumerate(lines) if l.startswith("@render "))
    j = next(k for k in range(i, len(lines)) if lines[k] == "@end")
    head = lines[i].split()
    return "\n".join(lines[i + 1:j]) + "\n", int(head[3]), int(head[5])
