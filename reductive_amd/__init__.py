"""reductive_amd -- MI355X-native PQ/OPQ encode-reconstruct path behind reductive's `Pq` surface.

Only the hot path lives here: `Pq.quantize_batch` / `Pq.reconstruct_batch` dispatch to
hand-written gfx950 kernels in libpqhip.so through the C ABI of include/pqhip.h.
"""
from ._lib import PqHipError, build, lib, lib_path  # noqa: F401
from . import qmatrix  # noqa: F401
from .pq import Pq, PanicError, ReductiveError, cluster_assignments, kmeans_iterations, train_pq, at_dot_b, opq_train_step, train_opq, bucket_eigenvalues, create_projection_matrix, rotate, set_rotation_variant, train_gaussian_opq, set_option, launch_log, vor2_tables  # noqa: F401

__all__ = ["Pq", "PanicError", "PqHipError", "ReductiveError", "build", "cluster_assignments",
           "kmeans_iterations", "train_pq", "at_dot_b", "opq_train_step", "train_opq", "bucket_eigenvalues", "create_projection_matrix", "rotate", "set_rotation_variant",
           "train_gaussian_opq", "set_option", "launch_log",
           "lib", "lib_path", "qmatrix"]
