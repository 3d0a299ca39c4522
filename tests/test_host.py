"""CPU tests of the host logic: the `Pq` mirror's panics, the C-ABI library's exports, and the
row-sharding arithmetic used for multi-GPU runs (gloo, world_size 2).  No GPU compute here."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ra():
    import reductive_amd
    if not os.path.exists(reductive_amd.lib_path()):
        reductive_amd.build()
    return reductive_amd


def test_library_loads_and_exports_every_declared_symbol(ra):
    L = ra.lib()
    hdr = open(os.path.join(ROOT, "include", "pqhip.h")).read()
    declared = set(re.findall(r"\b(pqhip_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"pqhip_status"}
    from reductive_amd._lib import EXPORTS
    assert declared == set(EXPORTS), declared ^ set(EXPORTS)
    for name in EXPORTS:
        assert hasattr(L, name), name
    assert L.pqhip_version() == 100
    assert L.pqhip_strerror(5).decode() == "no usable HIP device"


def test_no_oracle_in_product():
    """The product path must not reach into oracle/ (test infrastructure)."""
    for dp, _, files in os.walk(os.path.join(ROOT, "reductive_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"(import|from|include|dlopen|CDLL|LoadLibrary)[^\n]*oracle", src), f
    so = os.path.join(ROOT, "reductive_amd", "libpqhip.so")
    out = subprocess.run(["ldd", so], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_pq_new_panics(ra):
    with pytest.raises(ra.PanicError, match="without quantizers"):
        ra.Pq(None, np.zeros((0, 4, 3), np.float32))
    with pytest.raises(ra.PanicError, match="Incorrect projection matrix shape"):
        ra.Pq(np.eye(5, dtype=np.float32), np.zeros((2, 4, 3), np.float32))
    pq = ra.Pq(np.eye(6, dtype=np.float32), synth.normalish(1, (2, 4, 3)))
    assert pq.quantized_len() == 2 and pq.reconstructed_len() == 6
    assert pq.n_quantizer_centroids() == 4 and pq.projection().shape == (6, 6)
    assert pq == ra.Pq(np.eye(6, dtype=np.float32), synth.normalish(1, (2, 4, 3)))
    assert pq != ra.Pq(None, synth.normalish(1, (2, 4, 3)))


def test_shape_panics_happen_before_any_device_call(ra):
    pq = ra.Pq(None, synth.normalish(2, (2, 4, 3)))
    with pytest.raises(ra.PanicError, match="Quantizer and vector length mismatch"):
        pq.quantize_batch(np.zeros((3, 5), np.float32))
    with pytest.raises(ra.PanicError, match="Quantized matrix has incorrect shape"):
        pq.quantize_batch_into(np.zeros((3, 6), np.float32), np.zeros((3, 3), np.uint8))
    with pytest.raises(ra.PanicError, match="Reconstructions matrix has incorrect shape"):
        pq.reconstruct_batch_into(np.zeros((3, 2), np.uint8), np.zeros((3, 5), np.float32))
    with pytest.raises(ra.PanicError, match="Quantization length"):
        pq.reconstruct_batch(np.zeros((3, 3), np.uint8))
    with pytest.raises(ra.PanicError, match="Quantizer and vector length mismatch"):
        pq.quantize_vector(np.zeros(5, np.float32))


def test_single_vector_path_kats(ra, kats):
    k = kats["pq_predefined_codebook"]
    pq = ra.Pq(None, np.array(k["quantizers"], np.float32))
    for v, want, rec in zip(k["vectors"], k["quantizations"], k["reconstructions"]):
        assert pq.quantize_vector(np.array(v, np.float32), dtype=np.uint64).tolist() == want
        assert pq.reconstruct(np.array(want)).tolist() == rec
    w = kats["index_width"]
    ok = ra.Pq(None, synth.uniform01(3, (1, w["k_ok_u8"], w["dsub"])))
    ok.quantize_vector(synth.uniform01(4, (w["dsub"],)), dtype=np.uint8)          # pq.rs:442-450
    narrow = ra.Pq(None, synth.uniform01(5, (1, w["k_panic_u8"], w["dsub"])))
    with pytest.raises(ra.PanicError, match="Cannot store centroids"):           # pq.rs:452-461
        narrow.quantize_vector(synth.uniform01(6, (w["dsub"],)), dtype=np.uint8)


def test_batch_path_fails_loudly_without_a_gpu(ra):
    n = ctypes.c_int32(-1)
    rc = ra.lib().pqhip_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    pq = ra.Pq(None, synth.normalish(7, (2, 4, 3)))
    with pytest.raises(ra.PqHipError, match="no usable HIP device"):
        pq.quantize_batch(np.zeros((3, 6), np.float32))
    with pytest.raises(ra.PqHipError):
        pq.reconstruct_batch(np.zeros((3, 2), np.uint8))


def test_training_and_lookup_entry_points_fail_loudly_without_a_gpu(ra):
    """The "next" rows have no CPU fallback either; argument errors are reported before any device work."""
    n = ctypes.c_int32(-1)
    rc = ra.lib().pqhip_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    x = synth.normalish(8, (50, 6))
    with pytest.raises(ra.PqHipError, match="no usable HIP device"):
        ra.kmeans_iterations(synth.normalish(9, (2, 4, 3)), x, 2)
    with pytest.raises(ra.PqHipError, match="no usable HIP device"):
        ra.cluster_assignments(synth.normalish(9, (4, 6)), x)
    with pytest.raises(ra.PqHipError):
        ra.train_pq(2, 2, 3, 1, x)
    # host-side checks that mirror the reference's asserts come first
    with pytest.raises(ra.PanicError, match="Centroid and instance lengths differ"):
        ra.kmeans_iterations(synth.normalish(9, (2, 4, 3)), synth.normalish(8, (50, 7)), 1)
    with pytest.raises(ra.PanicError, match="zero centroids"):
        ra.kmeans_iterations(np.zeros((2, 0, 3), np.float32), x, 1)
    with pytest.raises(ra.ReductiveError, match="between 1 and 6, was 7"):
        ra.train_pq(7, 2, 3, 1, x)
    with pytest.raises(ra.ReductiveError, match="bits must be between 1 and 5"):
        ra.train_pq(2, 6, 3, 1, x)            # log2(50) truncates to 5 (pq.rs:76-81)
    with pytest.raises(ra.ReductiveError, match="iterations must be >= 1"):
        ra.train_pq(2, 2, 0, 1, x)


def test_bucket_eigenvalues_kats(ra, kats):
    """opq.rs:303-329: the eigenvalue allocation of the initial OPQ projection (host-side logic)."""
    k = kats["bucket_eigenvalues"]
    for c in k["cases"]:
        assert ra.bucket_eigenvalues(c["eigenvalues"], c["n_buckets"]) == c["expected"]
    with pytest.raises(ra.PanicError, match="multiple of the number of buckets"):
        ra.bucket_eigenvalues(k["uneven"]["eigenvalues"], k["uneven"]["n_buckets"])
    with pytest.raises(ra.PanicError, match="zero buckets"):
        ra.bucket_eigenvalues([1.0, 2.0], 0)
    with pytest.raises(ra.PanicError, match="positive eigenvalues"):
        ra.bucket_eigenvalues([-1.0, 2.0], 2)
    # linalg.rs:252-260: the covariance the initial projection is built from (host side, numpy)
    c = kats["covariance"]
    x = np.array(c["x"], np.float32)
    centered = x - x.mean(axis=0, dtype=np.float32)
    assert ((centered.T @ (centered / np.float32(len(x) - 1))).tolist()) == c["expected"]
    P = ra.create_projection_matrix(synth.normalish(77, (50, 6)), 3)
    assert np.abs(P.T @ P - np.eye(6, dtype=np.float32)).max() < 1e-5


@pytest.mark.parametrize("world,workload", [(2, "encode"), (8, "encode_d768")])
def test_bench_sharding_ranks_gloo(world, workload):
    """bench.py's N>1 path on CPU: world_size 2 and 8 over gloo, --dry-run (no GPU work): every rank takes
    its own shard, timing is max-reduced, rank 0 prints one JSON line; the 8-rank case is the BASELINE
    configs[4] workload (100 M x 768 over 8 GPUs = 12.5 M rows per rank; shrunk rows here)."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29531 + world), os.path.join(ROOT, "bench.py"),
           "--gpus", str(world), "--steps", "2", "--warmup", "1", "--dry-run", "--rows", "4096", "--workload", workload]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == world and rec["scaling"] == "weak" and rec["config"]["rows_total"] == 4096 * world
    assert rec["config"]["shards"] == [[r * 4096, (r + 1) * 4096] for r in range(world)]
    if workload == "encode_d768":
        assert rec["config"]["d"] == 768 and rec["config"]["M"] == 48 and "configs[4]" in rec["config"]["workload"]
