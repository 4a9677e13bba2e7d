"""MI355X-native implementation of SLIMM's alignment-to-profile hot path.

`slimm_amd.profiler.Slimm` mirrors the reference's `class slimm`; all compute is in libslimm_hip.so
(HIP kernels for gfx950 + C++ host glue) reached through the C ABI in include/slimm_hip.h.
"""
from .workload import Options, Records, Taxonomy, Workload  # noqa: F401

__all__ = ["Options", "Records", "Taxonomy", "Workload"]
