"""trueconsense_amd — MI355X-native pileup-tally + base-calling path of TrueConsense.

Host modules keep the reference's names (indexing, Events, Coverage, Sequences, Outputs,
TrueConsense); the arithmetic runs in hand-written HIP kernels for gfx950 behind the C ABI of
include/tcmi.h (trueconsense_amd/lib/libtcmi.so).  No CPU fallback exists.
"""
from .version import __version__  # noqa: F401
