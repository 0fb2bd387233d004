"""codon_amd: MI355X-native (gfx950) implementation of the CODON depth-super-resolution hot path.

    from codon_amd import CODONNet            # x4 / x8 form   (reference: CODON_X4/CODON_x4.py)
    from codon_amd import CODONNet16          # x16 form        (reference: CODON_X16/CODON_x16.py)

Drop-in module names for the reference's scripts (`from CODON_x4 import CODONNet`) live in
codon_amd/compat/: put that directory on sys.path.
"""
from .model import (BaseNet_RMCR_fuseRMCR, BaseNet_RMCR_fuseRMCR_cross, BasicConv, CAC_channel, CAC_spatial, ChannelGate, ChannelPool, CODONNet, CODONNet16,
                    Flatten, strip_module_prefix)

__all__ = ["CODONNet", "CODONNet16", "BaseNet_RMCR_fuseRMCR", "BaseNet_RMCR_fuseRMCR_cross", "CAC_channel", "CAC_spatial", "ChannelGate", "ChannelPool", "BasicConv",
           "Flatten", "strip_module_prefix"]
