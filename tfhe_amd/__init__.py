"""tfhe_amd — the bootstrapped-gate path of the TFHE crate on AMD MI355X.

The compute path is libtfhe_hip.so (hand-written HIP for gfx950 behind the C ABI
of include/tfhe_hip.h); this package is the loader and a thin host-side handle.
"""
from .engine import Engine, GATES, GATE_ID, TfheHipError, make_params  # noqa: F401
