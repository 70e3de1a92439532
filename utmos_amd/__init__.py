"""utmos_amd -- MI355X-native greedy maximum-coverage sample selection (the `utmos select` hot path).

Host code is plain Python + ctypes over libutmos_hip.so (hand-written HIP for gfx950); there is no
CPU fallback: without the library, or without a GPU, the selection entry points raise.
"""
__version__ = "2.2.0+mi355x.1"
