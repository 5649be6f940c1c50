"""fast_lio_amd -- MI355X-native measurement-update hot path of FAST-LIO2 (h_share_model + IEKF).

The product is the C-ABI library built from fast_lio_amd/csrc (see include/fastlio_hip.h); this
package holds the build driver, the ctypes binding used by tests/bench, the synthetic scene
generator and the multi-GPU glue.
"""
__version__ = "0.1.0"
