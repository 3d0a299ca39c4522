"""ctypes binding of libpqhip.so (include/pqhip.h).  Plumbing only -- no compute here."""
import ctypes
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("PQHIP_LIB") or os.path.join(_HERE, "libpqhip.so")   # PQHIP_LIB: A/B of two builds on one box (tools/)
_lib = None

OK, EINVAL, ESHAPE, ECODE_RANGE, EINDEX_WIDTH, ENODEV, EHIP, ENOMEM, EUNSUPPORTED = range(9)


class PqHipError(RuntimeError):
    def __init__(self, status, what=""):
        self.status = status
        msg = "status %d" % status
        if _lib is not None:
            msg = _lib.pqhip_strerror(status).decode()
            if status == EHIP:
                msg += " [" + _lib.pqhip_last_hip_error().decode() + "]"
        super().__init__("pqhip: %s%s" % (msg, (" (" + what + ")") if what else ""))


def lib_path():
    return _SO


def build(force=False):
    """Compile libpqhip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    srcdir = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", srcdir, "-s", "-j8"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    return _SO


def lib():
    """Load libpqhip.so.  Fails loudly if it is missing: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise ImportError("reductive_amd: %s is missing; run reductive_amd.build() "
                          "(needs hipcc). There is no CPU fallback." % _SO)
    # torch bundles its own libamdhip64 (same SONAME).  If torch is going to live in this
    # process it must be loaded first so that both share ONE HIP runtime.
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is optional for the C ABI itself
            pass
    L = ctypes.CDLL(_SO)
    i32, i64, vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p
    fp = ctypes.POINTER(ctypes.c_float)
    L.pqhip_version.restype = i32
    L.pqhip_strerror.restype = ctypes.c_char_p
    L.pqhip_strerror.argtypes = [i32]
    L.pqhip_last_hip_error.restype = ctypes.c_char_p
    L.pqhip_device_count.restype = i32
    L.pqhip_device_count.argtypes = [ctypes.POINTER(i32)]
    L.pqhip_ctx_create.restype = i32
    L.pqhip_ctx_create.argtypes = [ctypes.POINTER(i32), i32, ctypes.POINTER(vp)]
    L.pqhip_ctx_destroy.restype = None
    L.pqhip_ctx_destroy.argtypes = [vp]
    L.pqhip_ctx_n_devices.restype = i32
    L.pqhip_ctx_n_devices.argtypes = [vp]
    L.pqhip_codebook_create.restype = i32
    L.pqhip_codebook_create.argtypes = [vp, fp, i64, i64, i64, fp, ctypes.POINTER(vp)]
    L.pqhip_codebook_destroy.restype = None
    L.pqhip_codebook_destroy.argtypes = [vp]
    for name in ("quantized_len", "reconstructed_len", "n_centroids"):
        f = getattr(L, "pqhip_codebook_" + name)
        f.restype = i64
        f.argtypes = [vp]
    L.pqhip_codebook_has_projection.restype = i32
    L.pqhip_codebook_has_projection.argtypes = [vp]
    L.pqhip_quantize_batch_f32.restype = i32
    L.pqhip_quantize_batch_f32.argtypes = [vp, vp, i64, i64, i64, vp, i32, i64, i64]
    L.pqhip_reconstruct_batch_f32.restype = i32
    L.pqhip_reconstruct_batch_f32.argtypes = [vp, vp, i32, i64, i64, i64, vp, i64, i64]
    L.pqhip_quantize_batch_f32_dev.restype = i32
    L.pqhip_quantize_batch_f32_dev.argtypes = [vp, i32, vp, i64, i64, vp, i32, i64, vp]
    L.pqhip_reconstruct_batch_f32_dev.restype = i32
    L.pqhip_reconstruct_batch_f32_dev.argtypes = [vp, i32, vp, i32, i64, i64, vp, i64, vp]
    L.pqhip_reconstruct_rows_f32_dev.restype = i32
    L.pqhip_reconstruct_rows_f32_dev.argtypes = [vp, i32, vp, i32, i64, i64, vp, i64, vp, vp, i64, vp]
    L.pqhip_reconstruct_rows_records_f32_dev.restype = i32
    L.pqhip_reconstruct_rows_records_f32_dev.argtypes = [vp, i32, vp, i32, i64, i64, i64, vp, i64, vp, i64, vp]
    L.pqhip_adc_tables_f32_dev.restype = i32
    L.pqhip_adc_tables_f32_dev.argtypes = [vp, i32, vp, i64, i64, vp, vp]
    L.pqhip_adc_scan_f32_dev.restype = i32
    L.pqhip_adc_scan_f32_dev.argtypes = [vp, i32, vp, i64, vp, i32, i64, i64, vp, i64, vp]
    L.pqhip_check_codes_dev.restype = i32
    L.pqhip_check_codes_dev.argtypes = [vp, i32, vp]
    L.pqhip_cluster_assignments_f32.restype = i32
    L.pqhip_cluster_assignments_f32.argtypes = [vp, fp, i64, i64, vp, i64, i64, i64, vp, i32]
    L.pqhip_kmeans_iterations_f32.restype = i32
    L.pqhip_kmeans_iterations_f32.argtypes = [vp, fp, i64, i64, i64, vp, i64, i64, i64, i32, fp]
    L.pqhip_kmeans_iterations_f32_dev.restype = i32
    L.pqhip_kmeans_iterations_f32_dev.argtypes = [vp, i32, fp, i64, i64, i64, vp, i64, i64, i32, fp, vp]
    L.pqhip_opq_train_step_f32_dev.restype = i32
    L.pqhip_opq_train_step_f32_dev.argtypes = [vp, i32, fp, i64, i64, i64, fp, vp, i64, i64, fp, vp]
    L.pqhip_at_dot_b_f32_dev.restype = i32
    L.pqhip_at_dot_b_f32_dev.argtypes = [vp, i32, vp, i64, i64, vp, i64, i64, i64, fp, vp]
    L.pqhip_rotate_f32_dev.restype = i32
    L.pqhip_rotate_f32_dev.argtypes = [vp, i32, vp, i64, i64, i64, fp, vp, i64, vp]
    L.pqhip_matrix_upload_f32.restype = i32
    L.pqhip_matrix_upload_f32.argtypes = [vp, i32, vp, i64, i64, i64, i64, ctypes.POINTER(vp)]
    L.pqhip_matrix_device_ptr.restype = vp
    L.pqhip_matrix_device_ptr.argtypes = [vp]
    L.pqhip_matrix_rows.restype = i64
    L.pqhip_matrix_rows.argtypes = [vp]
    L.pqhip_matrix_destroy.restype = None
    L.pqhip_matrix_destroy.argtypes = [vp]
    L.pqhip_set_encode_variant.restype = i32
    L.pqhip_set_encode_variant.argtypes = [vp, i32]
    L.pqhip_set_rotation_variant.restype = i32
    L.pqhip_set_rotation_variant.argtypes = [i32]
    L.pqhip_last_encode_kernel.restype = ctypes.c_char_p
    L.pqhip_last_encode_kernel.argtypes = [vp]
    L.pqhip_ctx_set_option.restype = i32
    L.pqhip_ctx_set_option.argtypes = [vp, ctypes.c_char_p, i64]
    L.pqhip_launch_log.restype = ctypes.c_char_p
    L.pqhip_launch_log.argtypes = []
    L.pqhip_launch_log_reset.restype = None
    L.pqhip_launch_log_reset.argtypes = []
    L.pqhip_vor2_tables_host.restype = i32
    L.pqhip_vor2_tables_host.argtypes = [vp, i64, i64, i64, vp, i64, vp, ctypes.POINTER(i64)]
    L.pqhip_selftest_mfma_chain.restype = i32
    L.pqhip_selftest_mfma_chain.argtypes = [vp, i32, i32, i32, ctypes.c_uint64,
                                            ctypes.POINTER(i64)]
    _lib = L
    return L


# every symbol include/pqhip.h declares (checked by the CPU test-suite)
EXPORTS = [
    "pqhip_version", "pqhip_strerror", "pqhip_last_hip_error", "pqhip_device_count",
    "pqhip_ctx_create", "pqhip_ctx_destroy", "pqhip_ctx_n_devices", "pqhip_codebook_create",
    "pqhip_codebook_destroy", "pqhip_codebook_quantized_len",
    "pqhip_codebook_reconstructed_len", "pqhip_codebook_n_centroids",
    "pqhip_codebook_has_projection", "pqhip_quantize_batch_f32", "pqhip_reconstruct_batch_f32",
    "pqhip_quantize_batch_f32_dev", "pqhip_reconstruct_batch_f32_dev", "pqhip_reconstruct_rows_f32_dev", "pqhip_reconstruct_rows_records_f32_dev", "pqhip_check_codes_dev",
    "pqhip_adc_tables_f32_dev", "pqhip_adc_scan_f32_dev",
    "pqhip_cluster_assignments_f32", "pqhip_kmeans_iterations_f32", "pqhip_kmeans_iterations_f32_dev",
    "pqhip_opq_train_step_f32_dev", "pqhip_at_dot_b_f32_dev", "pqhip_rotate_f32_dev",
    "pqhip_matrix_upload_f32", "pqhip_matrix_device_ptr", "pqhip_matrix_rows", "pqhip_matrix_destroy",
    "pqhip_set_encode_variant", "pqhip_set_rotation_variant", "pqhip_last_encode_kernel", "pqhip_selftest_mfma_chain",
    "pqhip_ctx_set_option", "pqhip_launch_log", "pqhip_launch_log_reset", "pqhip_vor2_tables_host",
]
