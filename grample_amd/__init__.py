"""grample_amd: MI355X-native Gibbs sampling hot path for grample (UAI Markov networks).

The product is the C-ABI shared library grample_amd/libgrample_hip.so
(include/grample_hip.h); this package is its ctypes view for tests and bench.
"""
from . import _lib  # noqa: F401
from .api import (AdaptSelect, ChainConvergence, GR_FLAG_CHAIN_PER_LANE, GR_FLAG_DIRECT, GR_FLAG_SCAN_PER_CHAIN, GR_FLAG_SCAN_SYSTEMATIC, GR_FLAG_SEQUENTIAL, GR_FLAG_STATE_GLOBAL, GrError, HIPChains, MergeChains, Model, NewErrorSuite,  # noqa: F401
                  Solution, allreduce_counts, device_count, philox4x32_10, rccl_calls, split)
