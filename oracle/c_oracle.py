"""ctypes wrapper of the plain-C oracle (oracle/codon_oracle.c) -- TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(_HERE, "libcodon_oracle.so")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, "codon_oracle.c")):
            subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
        _lib = C.CDLL(so)
        _lib.codon_oracle_forward.restype = C.c_int
        _lib.codon_oracle_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    return _lib


def flat_params(sd) -> np.ndarray:
    """state_dict (any variant) -> the flat buffer codon_oracle.c expects (49- or 44-tensor order, used tensors only)."""
    from .codon_oracle import CONV_SHAPES
    parts = [np.asarray(sd[k], dtype=np.float32).ravel() for k, _ in CONV_SHAPES]
    for i in range(5):
        for k in ("mlp.1.weight", "mlp.1.bias", "mlp.3.weight", "mlp.3.bias"):
            parts.append(np.asarray(sd[f"attention_c{i}.{k}"], dtype=np.float32).ravel())
    for i in range(5):
        parts.append(np.asarray(sd[f"attention_s{i}.spatial.conv.weight"], dtype=np.float32).ravel())
    return np.ascontiguousarray(np.concatenate(parts))


def forward(sd, x, y) -> np.ndarray:
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
    y = np.ascontiguousarray(np.asarray(y, dtype=np.float32))
    B, _, H, W = x.shape
    p = flat_params(sd)
    out = np.empty_like(x)
    rc = lib().codon_oracle_forward(p.ctypes.data, x.ctypes.data, y.ctypes.data, out.ctypes.data, B, H, W)
    if rc != 0:
        raise MemoryError("codon_oracle_forward failed")
    return out
