"""Drop-in for the reference's CODON_X4/attention/ResCBAM.py (:6-37, :67-81): only the classes a pickled
CODONNet can name.  `attention_c5` is a ChannelGate whose mlp[0] is this module's Flatten; both are state only
(never executed on the CODONNet path)."""
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
if _root not in _sys.path:
    _sys.path.insert(0, _root)

from codon_amd.model import BasicConv, ChannelGate, ChannelPool, Flatten  # noqa: E402,F401
from codon_amd.model import CAC_spatial as SpatialGate  # noqa: E402,F401  (same state layout: compress + spatial.conv)
