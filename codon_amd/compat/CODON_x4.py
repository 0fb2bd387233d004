"""Drop-in for the reference's CODON_X4/CODON_x4.py: `from CODON_x4 import CODONNet`
(/root/reference/CODON_X4/test.py:12,48)."""
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _root not in _sys.path:
    _sys.path.insert(0, _root)

from codon_amd.model import CODONNet  # noqa: E402,F401
from codon_amd.model import CAC_channel as CHANNEL, CAC_spatial as SPATIAL, ChannelGate  # noqa: E402,F401
