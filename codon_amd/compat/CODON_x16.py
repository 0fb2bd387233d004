"""Drop-in for the reference's CODON_X16/CODON_x16.py: `from CODON_x16 import CODONNet`
(/root/reference/CODON_X16/test.py:15,51).  The x16 class has no attention_c5/attention_s5."""
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _root not in _sys.path:
    _sys.path.insert(0, _root)

from codon_amd.model import CODONNet16 as CODONNet  # noqa: E402,F401
from codon_amd.model import CAC_channel as CHANNEL, CAC_spatial as SPATIAL  # noqa: E402,F401
from codon_amd.model import BaseNet_RMCR_fuseRMCR  # noqa: E402,F401  (ablation class of the same file, :16-90)
