"""dg_tta_amd — MI355X-native engine for the DG-TTA test-time-adaptation hot path.

Mirrors the reference package layout for that path (dg_tta/{gin,mind,utils,run}.py, dg_tta/tta/*.py); every
numeric op is executed by hand-written gfx950 kernels behind the C ABI in include/dgtta.h.  There is no CPU
fallback: ops raise `DgttaError` when libdgtta_hip.so is missing or a CPU tensor is passed.
"""
__version__ = "0.1.0"
