"""Package shim: the reference's x4/x8 networks import `attention.ResCBAM` (CODON_X4/CODON_x4.py:6), so whole-module
pickles of them name `attention.ResCBAM.ChannelGate` / `.Flatten` (attention_c5, CODON_x4.py:64)."""
