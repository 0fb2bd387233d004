"""Drop-in for the reference's CAC_module.py (/root/reference/CODON_X4/CAC_module.py): needed to
unpickle whole-module checkpoints ({"model": <nn.Module>}, test.py:56-59), which name these classes."""
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _root not in _sys.path:
    _sys.path.insert(0, _root)

from codon_amd.model import BasicConv, CAC_channel, CAC_spatial, ChannelPool, Flatten  # noqa: E402,F401
