"""Image / checkpoint plumbing of the reference script, host side only:
   PNG -> grey uint8 -> /255 -> (1,1,H,W) float32      /root/reference/CODON_X4/test.py:116-123
   {"epoch", "model": <nn.Module>} pickles and 'module.'-prefixed state dicts      test.py:56-59, CODON_X16/test.py:52-60

Grey conversion: the reference uses cv2.imread(name, 0); OpenCV is absent here, PIL's convert('L') is used
instead (ITU-R 601 weights, may differ from OpenCV by +-1 on colour images) -- PARITY UNPINNED for RGB inputs;
single-channel PNGs (all depth maps and labels) are read identically."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch


def read_gray(path: str) -> np.ndarray:
    from PIL import Image
    return np.asarray(Image.open(path).convert("L"), dtype=np.uint8)


def write_gray(path: str, img_u8: np.ndarray):
    from PIL import Image
    Image.fromarray(np.asarray(img_u8, dtype=np.uint8), mode="L").save(path)


def to_input(pic_u8: np.ndarray) -> torch.Tensor:
    """torch.from_numpy(pic / 255).float().unsqueeze(0).unsqueeze(0)   (float64 divide, then float32)."""
    return torch.from_numpy(np.asarray(pic_u8) / 255).float().unsqueeze(0).unsqueeze(0)


def _compat_on_path():
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "compat")
    if d not in sys.path:
        sys.path.insert(0, d)


def load_checkpoint(path: str, model: torch.nn.Module, strict: bool = True) -> int:
    """Load the reference's checkpoint formats into `model`; returns the stored epoch (or -1).
    Whole-module pickles name the classes CODON_x4.CODONNet / CAC_module.*: codon_amd/compat provides them."""
    from .model import strip_module_prefix
    _compat_on_path()
    ck = torch.load(path, map_location="cpu", weights_only=False)
    epoch = -1
    if isinstance(ck, dict) and "model" in ck:
        epoch = int(ck.get("epoch", -1))
        ck = ck["model"]
    sd = ck.state_dict() if isinstance(ck, torch.nn.Module) else ck
    model.load_state_dict(strip_module_prefix(sd), strict=strict)
    return epoch
