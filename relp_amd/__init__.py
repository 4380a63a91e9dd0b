"""relp_amd: MI355X-native revised-simplex hot path behind relp's trait surface.

The package holds the HIP kernels + C ABI (``csrc/``, built into ``librelp_amd.so``) and a thin ctypes binding
(``api``).  Nothing here computes on the CPU: without the built library and a HIP device every solve call raises.
"""
from .api import (  # noqa: F401
    Model, Solver, Batch, Options, Result, RelpError, default_options, lib, LIB_PATH, SYMBOLS,
    FINITE_OPTIMUM, INFEASIBLE, UNBOUNDED, ITERATION_LIMIT,
    STEEPEST_EDGE, DANTZIG, FIRST_PROFITABLE, FIRST_PROFITABLE_MEMORY,
    STOP_NO_ENTERING, STOP_UNBOUNDED, STOP_BUDGET,
)
