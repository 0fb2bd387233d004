"""ctypes binding of libcodon_hip.so (C ABI in include/codon_hip.h).

There is NO fallback: if the shared library is missing, was built for another ABI version or a
call fails, this module raises.  `import torch` happens before the CDLL so that the library's
NEEDED libamdhip64.so.7 resolves to the HIP runtime torch already loaded (one runtime per
process: streams and device pointers are shared with torch).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch  # noqa: F401  (must precede CDLL, see module docstring)

ABI_VERSION = 1
LIB_NAME = "libcodon_hip.so"

OK = 0
F32, BF16, F16 = 0, 1, 2
CONV_RELU, CONV_ADD_RESIDUAL, CONV_ACCUM_OUT, CONV_MASK_RELU, CONV_F16X3, CONV_MASK_SUM = 1, 2, 4, 8, 16, 32
PACK_FWD, PACK_DGRAD, PACK_FWD_F16X3, PACK_CHAIN1X1, PACK_CHAIN1X1_F16X3 = 0, 1, 2, 3, 4
TILING_8X32, TILING_4X32, TILING_4X32_SOLO, TILING_2X32_COUT_SPLIT = 0, 1, 2, 3      # codon_conv_tiling_f32
CAC_FOLDS = 16   # CODON_CAC_FOLDS


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "batch", "height", "width", "cin", "cout", "ksize", "x_ctotal", "x_coff", "y_ctotal", "y_coff",
        "r_ctotal", "r_coff", "flags", "dtype")]


class Tensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("ctotal", C.c_int32), ("coff", C.c_int32)]


WSUM_MAX = 32   # CODON_WSUM_MAX


class WsumDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("reserved", C.c_int32), ("data", C.c_void_p * WSUM_MAX), ("bytes", C.c_uint64 * WSUM_MAX)]


REDUCE_MAX_USES, REDUCE_MAX_ITEMS = 5, 44          # CODON_REDUCE_MAX_USES / _ITEMS
REDUCE_ACCUMULATE, REDUCE_WGRAD, REDUCE_FLIP9 = 1, 2, 4
WGRAD_DEFER = 2                                     # CODON_WGRAD_DEFER
W1_FLIP, W1_ACCUMULATE, W1_DEFER = 1, 2, 4          # CODON_W1_*


CAST_MAX = 32                                       # CODON_CAST_MAX


ADAM_MAX = 64                                       # CODON_ADAM_MAX


class AdamDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("reserved", C.c_int32), ("param", C.c_void_p * ADAM_MAX), ("count", C.c_int64 * ADAM_MAX)]


class CastDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("reserved", C.c_int32), ("src", C.c_void_p * CAST_MAX), ("count", C.c_int64 * CAST_MAX),
                ("dtype", C.c_int32 * CAST_MAX)]


class ReduceItem(C.Structure):
    _fields_ = [("out", C.c_void_p), ("part", C.c_void_p * REDUCE_MAX_USES), ("stride", C.c_int64), ("nuse", C.c_int32),
                ("nparts", C.c_int32), ("cout", C.c_int32), ("cin", C.c_int32), ("taps", C.c_int32), ("nchunk", C.c_int32),
                ("flags", C.c_int32), ("reserved", C.c_int32)]


_P, _I, _S = C.c_void_p, C.c_int32, C.c_size_t
_TP = C.POINTER(Tensor)

# name -> (restype, argtypes); mirrors include/codon_hip.h one to one
SIGNATURES = {
    "codon_abi_version": (C.c_int, []),
    "codon_last_error_string": (C.c_char_p, []),
    "codon_build_source_hash": (C.c_char_p, []),
    "codon_conv_packed_weight_bytes": (_S, [_I, _I, _I, _I]),
    "codon_conv_pack_weight": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "codon_conv2d_fwd": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P]),
    "codon_conv2d_sum_into_fwd": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P]),
    "codon_conv_chain1x1_fwd": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _TP, _TP, _P]),
    "codon_conv_chain1x1_stats_fwd": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _TP, _TP, _P, _P, _I, _P]),
    "codon_cac_fused_tiles": (_I, [_I, _I]),
    "codon_cac_fused_parts": (_I, [_I, _I, _I]),
    "codon_cac_fused_finish": (C.c_int, [_I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "codon_cac_gate_folded_fwd": (C.c_int, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "codon_cac_tail_fwd": (C.c_int, [_I, _I, _I, _I] + [_P] * 14 + [_P]),
    "codon_conv2d_gated_fwd": (C.c_int, [C.POINTER(ConvDesc), _P, _TP, _P, _P, _P, _P, _P]),
    "codon_conv2d_gated_emit_fwd": (C.c_int, [C.POINTER(ConvDesc), _P, _TP, _P, _P, _P, _P, _TP, _P]),
    "codon_conv_wgrad_workspace_bytes": (_S, [C.POINTER(ConvDesc)]),
    "codon_conv2d_wgrad": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _S, _I, _P]),
    "codon_conv1x1_bwd": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _TP, _P, _P, C.c_size_t, _I, _P]),
    "codon_stem_fwd": (C.c_int, [_I, _I, _I, _P, _P, _P, _I, _I, _I, _P]),
    "codon_stem_pair_fwd": (C.c_int, [_I, _I, _I, _P, _P, _P, _I, _I, _P, _P, _P, _I, _I, _I, _P]),
    "codon_head_fwd": (C.c_int, [_I, _I, _I, _P, _I, _I, _P, _P, _P, _I, _P]),
    "codon_head_fwd_y16": (C.c_int, [_I, _I, _I, _P, _I, _I, _P, _P, _P, _I, _P]),
    "codon_cac_stats_tiles": (_I, [_I, _I]),
    "codon_cac_stats_fwd": (C.c_int, [_I, _I, _I, _TP, _TP, _P, _P, _I, _P]),
    "codon_cac_stats_scaled_fwd": (C.c_int, [_I, _I, _I, _TP, _TP, _P, _P, _P, _I, _P]),
    "codon_ew_sq_scale": (C.c_int, [_I, _I, _I, _TP, _P, _TP, _I, _P]),
    "codon_cac_gate_fwd": (C.c_int, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "codon_cac_spatial_fwd": (C.c_int, [_I, _I, _I, _P, _P, _P, _P]),
    "codon_stencil_1to64": (C.c_int, [_I, _I, _I, _P, _P, _TP, _I, _TP, _I, _P]),
    "codon_conv1ch_wgrad_workspace_bytes": (_S, [_I, _I, _I]),
    "codon_conv1ch_wgrad": (C.c_int, [_I, _I, _I, _TP, _P, _P, _I, _P, _S, _I, _P]),
    "codon_ew_add_mask": (C.c_int, [_I, _I, _I, _I, _TP, _TP, _TP, _I, _I, _P]),
    "codon_cac_bwd_tiles": (_I, [_I, _I]),
    "codon_cac_bwd_spatial_blocks": (_I, [_I, _I, _I]),
    "codon_cac_bwd_reduce": (C.c_int, [_I, _I, _I, _TP, _TP, _TP, _TP, _P, _P, _P, _P, _P, _P, _I, _P]),
    "codon_cac_bwd_gate": (C.c_int, [_I, _I, _I] + [_P] * 14 + [_P]),
    "codon_cac_bwd_spatial": (C.c_int, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "codon_cac_bwd_apply": (C.c_int, [_I, _I, _I, _TP, _TP, _TP, _TP, _P, _P, _P, _P, _P, _P, _TP, _TP, _TP, _TP,
                                      _I, _I, _P]),
    "codon_postprocess_u8": (C.c_int, [C.c_int64, _P, _P, _P]),
    "codon_postprocess_u8_dt": (C.c_int, [C.c_int64, _P, C.c_int32, _P, _P]),
    "codon_masked_sqerr": (C.c_int, [C.c_int64, _P, _P, _P, _P]),
    "codon_ssim_tiles": (_I, [_I, _I, _I]),
    "codon_ssim_fwd": (C.c_int, [_I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "codon_l1_fwd": (C.c_int, [C.c_int64, _P, _P, _P, _I, _P, _P]),
    "codon_ssim_l1_bwd": (C.c_int, [_I, _I, _I, _P, _P, _P, _P, _P, C.c_float, C.c_float, _P]),
    "codon_bicubic_upsample": (C.c_int, [_I, _I, _I, _I, _P, _P, _P, _P]),
    "codon_cac_apply_fwd": (C.c_int, [_I, _I, _I, _TP, _TP, _P, _P, _TP, _TP, _TP, _TP, _I, _P]),
    "codon_ew_sum_mask": (C.c_int, [_I, _I, _I, _I, _TP, _I, _TP, _TP, _TP, _TP, _TP, _I, _P]),
    "codon_conv1x1_bwd_gated": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _TP, _P, _P, _S, _I, _P, _P, _P, _P, _P, _P, _I, _P]),
    "codon_cac_bwd_reduce_acc": (C.c_int, [_I, _I, _I, _TP, _TP, _TP, _TP, _P, _P, _P, _P, _P, _P, _P, _P, _TP, _TP, _I, _I, _P]),
    "codon_conv_pair_begin": (C.c_int, []),
    "codon_conv_pair_end": (C.c_int, [_P]),
    "codon_conv_tiling_f32": (C.c_int, [C.POINTER(ConvDesc), C.c_int, C.c_int]),
    "codon_cast_multi": (C.c_int, [C.POINTER(CastDesc), _P, _P]),
    "codon_adam_step": (C.c_int, [_P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _I, _P]),
    "codon_reduce_multi": (C.c_int, [C.POINTER(ReduceItem), _I, _P]),
    "codon_weight_checksum_workspace_bytes": (_S, []),
    "codon_weight_checksum": (C.c_int, [C.POINTER(WsumDesc), _P, _P, _I, _P, _P]),
}

_lock = threading.Lock()
_lib = None


_warned_override = False


def lib_path() -> str:
    """In-tree library; CODON_AMD_LIB names another build of it (kernel A/B timing, tools/ab_build.sh) -- said once
    on stderr, because it redirects the product library for every consumer in the process."""
    global _warned_override
    override = os.environ.get("CODON_AMD_LIB")
    if override and not _warned_override:
        _warned_override = True
        import sys
        print(f"codon_amd: CODON_AMD_LIB overrides the in-tree {LIB_NAME}: loading {override}", file=sys.stderr)
    return override or os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)


def load():
    """Load (once) and return the bound library; raise if it cannot be used."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError(
                f"codon_amd: {path} not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C codon_amd/csrc`. There is no CPU or eager fallback.")
        # PyDLL: the entry points are called WITH the interpreter lock held.  Every one of them is an asynchronous launch (or
        # host arithmetic) of a few microseconds; releasing and re-taking the lock around each -- what CDLL does -- hands it
        # to any other thread that wants it ~50 times per forward, and the launching thread then queues for it behind that
        # thread's whole time slice (measured: the pipelined test loop of codon_amd.infer ran SLOWER than the serial one,
        # 111 vs 142 images/s, with CDLL).  torch's own blocking calls (synchronize, .item()) still release the lock.
        lib = (C.CDLL if os.environ.get("CODON_LIB_CDLL") == "1" else C.PyDLL)(path)     # CODON_LIB_CDLL=1: A/B of the binding mode
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        v = lib.codon_abi_version()
        if v != ABI_VERSION:
            raise RuntimeError(f"codon_amd: {path} has ABI version {v}, host code expects {ABI_VERSION}")
        _lib = lib
        return lib


def build_info() -> dict:
    """Stale-binary guard: the source hash embedded in the loaded .so vs the hash of the sources now in the tree
    (same recipe as codon_amd/csrc/Makefile: csrc/*.hip + csrc/*.h in byte-sorted name order, then
    include/codon_hip.h, contents concatenated, sha256, first 32 hex digits)."""
    import glob
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    csrc = os.path.join(here, "csrc")
    names = sorted(os.path.basename(f) for f in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")))
    files = [os.path.join(csrc, n) for n in names] + [os.path.join(os.path.dirname(here), "include", "codon_hip.h")]
    path = lib_path()
    info = {"path": path, "source_hash_built": load().codon_build_source_hash().decode(),
            "source_hash_now": None, "so_mtime": os.path.getmtime(path), "newest_source_mtime": None}
    if names and all(os.path.exists(f) for f in files):   # an installed package may ship the .so without its sources
        h = hashlib.sha256()
        for f in files:
            with open(f, "rb") as fh:
                h.update(fh.read())
        info["source_hash_now"] = h.hexdigest()[:32]
        info["newest_source_mtime"] = max(os.path.getmtime(f) for f in files)
    return info


def check(status: int, what: str):
    if status != OK:
        msg = load().codon_last_error_string().decode(errors="replace")
        raise RuntimeError(f"codon_amd: {what} failed with status {status}: {msg}")
