"""reductive_amd -- MI355X-native PQ/OPQ encode-reconstruct path behind reductive's `Pq` surface.

Only the hot path lives here: `Pq.quantize_batch` / `Pq.reconstruct_batch` dispatch to
hand-written gfx950 kernels in libpqhip.so through the C ABI of include/pqhip.h.
"""
from ._lib import PqHipError, build, lib, lib_path  # noqa: F401
from .pq import Pq, PanicError, cluster_assignments  # noqa: F401

__all__ = ["Pq", "PanicError", "PqHipError", "build", "cluster_assignments", "lib", "lib_path"]
