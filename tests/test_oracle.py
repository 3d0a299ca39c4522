"""CPU tests: pin oracle/pq_oracle.c against the reference's own KATs and against an
independent exact-rational model of CANON-F32 (tests/exact_f32.py)."""
import numpy as np
import pytest

import exact_f32 as ex
import synth
from oracle import pq_oracle as orc


# ---- reference KATs (tests/golden/reference_kats.json) -----------------------------------
def test_kat_pq_quantize_batch(kats):
    k = kats["pq_predefined_codebook"]
    q = np.array(k["quantizers"], np.float32)
    x = np.array(k["vectors"], np.float32)
    for dt in (np.uint8, np.uint16, np.uint32, np.uint64):
        codes = orc.quantize_batch(q, x, dtype=dt)
        assert codes.dtype == dt
        assert codes.tolist() == k["quantizations"]


def test_kat_pq_quantize_vector(kats):
    k = kats["pq_predefined_codebook"]
    q = np.array(k["quantizers"], np.float32)
    for v, want in zip(k["vectors"], k["quantizations"]):
        assert orc.quantize_vector(q, np.array(v, np.float32)).tolist() == want


def test_kat_pq_reconstruct(kats):
    k = kats["pq_predefined_codebook"]
    q = np.array(k["quantizers"], np.float32)
    codes = np.array(k["quantizations"], np.uint64)
    rec = orc.reconstruct_batch(q, codes)
    assert rec.tolist() == k["reconstructions"]
    rec8 = orc.reconstruct_batch(q, codes.astype(np.uint8))
    assert rec8.tolist() == k["reconstructions"]
    assert q.shape[0] == k["quantized_len"] and q.shape[0] * q.shape[2] == k["reconstructed_len"]


def test_kat_cluster_assignments(kats):
    k = kats["cluster_assignments"]
    c = np.array(k["centroids"], np.float32)
    x = np.array(k["instances"], np.float32)
    assert orc.cluster_assignments(c, x).tolist() == k["assignments"]
    # transposed storage of the same instances (kmeans.rs:397-399): same answer
    xt = np.asfortranarray(x)
    assert orc.quantize_batch(c[None], xt, dtype=np.uint64)[:, 0].tolist() == k["assignments"]


def test_kat_squared_distance(kats):
    k = kats["squared_euclidean_distance"]
    a, b = k["ix2_ix2"]["a"], k["ix2_ix2"]["b"]
    assert orc.sqdist(np.array(a), np.array(b)).tolist() == k["ix2_ix2"]["expected"]
    a1 = np.array([k["ix1_ix2"]["a"]])
    assert orc.sqdist(a1, np.array(k["ix1_ix2"]["b"]))[0].tolist() == k["ix1_ix2"]["expected"]
    a0, b0 = np.array([k["ix1_ix1"]["a"]]), np.array([k["ix1_ix1"]["b"]])
    assert orc.sqdist(a0, b0)[0, 0] == k["ix1_ix1"]["expected"]


def test_kat_index_width(kats):
    k = kats["index_width"]
    q = synth.uniform01(1, (1, k["k_ok_u8"], k["dsub"]))
    x = synth.uniform01(2, (3, k["dsub"]))
    codes = orc.quantize_batch(q, x)           # K=256 fits u8
    assert codes.dtype == np.uint8 and codes.shape == (3, 1)


# ---- exact-rational cross-check of the declared arithmetic --------------------------------
@pytest.mark.parametrize("n", [1, 3, 7, 8, 9, 16, 20, 23, 31])
def test_dot_unrolled_matches_exact(n):
    x = synth.normalish(10 + n, (n,))
    y = synth.normalish(50 + n, (n,))
    assert float(orc.dot_unrolled(x, y)) == ex.dot_unrolled(list(x), list(y))


@pytest.mark.parametrize("shape", [(5, 2, 4, 20), (4, 3, 8, 16), (3, 5, 5, 3), (2, 1, 3, 7)])
def test_quantize_matches_exact(shape):
    n, M, K, dsub = shape
    q = synth.normalish(3, (M, K, dsub))
    x = synth.normalish(4, (n, M * dsub))
    assert orc.quantize_batch(q, x, dtype=np.uint64).tolist() == ex.quantize_batch(q, x).tolist()
    d_c = orc.sqdist(x[:, :dsub], q[0])
    d_e = ex.sqdist(x[:, :dsub], q[0])
    assert d_c.tobytes() == d_e.tobytes()


def test_rotation_kc_split_matches_exact():
    d = 260                                     # > KC=256 -> two k-blocks
    x = synth.normalish(5, (2, d))
    P = synth.normalish(6, (d, d))[:, :]        # arbitrary matrix is enough for the arithmetic
    got = orc.rotate(x, P)
    want = ex.rotate(x, P)
    assert got.tobytes() == want.tobytes()


def test_opq_quantize_reconstruct_matches_exact():
    M, K, dsub = 2, 4, 3
    d = M * dsub
    q = synth.normalish(7, (M, K, dsub))
    x = synth.normalish(8, (4, d))
    P = synth.orthonormal(9, d)
    codes = orc.quantize_batch(q, x, projection=P, dtype=np.uint64)
    assert codes.tolist() == ex.quantize_batch(q, x, projection=P).tolist()
    rec = orc.reconstruct_batch(q, codes, projection=P)
    assert rec.tobytes() == ex.reconstruct_batch(q, codes, projection=P).tobytes()


# ---- tie-break / special values (not pinned by the reference's tests; declared) -----------
def test_exact_tie_lowest_index_wins():
    q = np.zeros((1, 6, 4), np.float32)
    q[0, 1] = q[0, 4] = [1, 2, 3, 4]            # duplicates: 1 and 4
    q[0, 2] = q[0, 3] = [9, 9, 9, 9]
    q[0, 0] = [5, 5, 5, 5]
    q[0, 5] = [-1, 0, 0, 0]
    x = np.array([[1, 2, 3, 4], [9, 9, 9, 9], [0, 0, 0, 0]], np.float32)
    assert orc.quantize_batch(q, x)[:, 0].tolist() == [1, 2, 5]


def test_nan_and_inf_rows():
    q = synth.normalish(11, (2, 8, 4))
    x = synth.normalish(12, (5, 8))
    x[0, 0] = np.nan                            # whole first sub-row distance is NaN -> index 0
    x[1, 5] = np.inf                            # xx = inf, dp = +-inf -> NaN or +inf per centroid
    x[2, :] = 0
    codes = orc.quantize_batch(q, x)
    assert codes[0, 0] == 0                     # all-NaN row of distances -> index 0
    assert codes.tolist() == ex.quantize_batch(q, x).tolist()
    # a NaN centroid is never chosen while a finite distance exists
    q2 = q.copy()
    q2[0, 0, 0] = np.nan
    c2 = orc.quantize_batch(q2, x[2:])
    assert (c2[:, 0] != 0).all()
    assert c2.tolist() == ex.quantize_batch(q2, x[2:]).tolist()


def test_first_min_total_order():
    nan = float("nan")
    assert orc.first_min(np.array([nan, nan], np.float32)) == 0
    assert orc.first_min(np.array([nan, 3, 1, 1], np.float32)) == 2
    assert orc.first_min(np.array([0.0, -0.0], np.float32)) == 0
    assert orc.first_min(np.array([np.inf, np.inf, -np.inf], np.float32)) == 2


def test_large_norm_rows_create_ties():
    # |x|^2 >> |c|^2: xx + cc rounds identically for many centroids -> rounding ties.
    q = synth.normalish(13, (1, 16, 4)) * np.float32(1e-3)
    x = synth.normalish(14, (6, 4)) * np.float32(1e4)
    assert orc.quantize_batch(q, x, dtype=np.uint64).tolist() == ex.quantize_batch(q, x).tolist()


def test_strided_inputs_and_outputs():
    q = synth.normalish(15, (3, 8, 4))
    big = synth.normalish(16, (10, 40))
    x = big[::2, 5:17]                          # row stride 80, unit col stride, offset
    want = orc.quantize_batch(q, np.ascontiguousarray(x))
    assert orc.quantize_batch(q, x).tolist() == want.tolist()
    assert orc.quantize_batch(q, np.asfortranarray(x)).tolist() == want.tolist()
    out = np.zeros((3, 5), np.uint8).T          # transposed output buffer
    orc.quantize_batch(q, x, out=out)
    assert out.tolist() == want.tolist()


def test_multithreaded_oracle_is_identical():
    q = synth.normalish(17, (15, 256, 20))
    x = synth.normalish(18, (257, 300))
    a = orc.quantize_batch(q, x, n_threads=1)
    b = orc.quantize_batch(q, x, n_threads=5)
    assert a.tobytes() == b.tobytes()


def test_reconstruct_rejects_out_of_range_code():
    q = synth.normalish(19, (2, 5, 3))
    with pytest.raises(IndexError):
        orc.reconstruct_batch(q, np.array([[0, 5]], np.uint8))


def test_empty_batch():
    q = synth.normalish(20, (2, 4, 3))
    assert orc.quantize_batch(q, np.zeros((0, 6), np.float32)).shape == (0, 2)
    assert orc.reconstruct_batch(q, np.zeros((0, 2), np.uint8)).shape == (0, 6)


def test_brute_force_f64_agrees_when_gap_is_clear():
    # independent formulation: argmin ||x-c||^2 in float64 wherever the gap is far above fp32 noise
    q = synth.normalish(21, (15, 256, 20))
    x = synth.normalish(22, (64, 300))
    codes = orc.quantize_batch(q, x)
    xs = x.reshape(64, 15, 20).astype(np.float64)
    d = ((xs[:, :, None, :] - q[None].astype(np.float64)) ** 2).sum(-1)      # [n, M, K]
    srt = np.sort(d, axis=-1)
    clear = (srt[..., 1] - srt[..., 0]) > 1e-3
    assert clear.mean() > 0.9
    assert (d.argmin(-1)[clear] == codes[clear]).all()


# ---- "next" row: the k-means step of training (kmeans.rs:166-198, 308-360) ----------------------
def test_kat_update_centroids(kats):
    k = kats["update_centroids"]
    x = np.array(k["instances"], np.float32)
    got = orc.update_centroids(k["centroids_shape"], x, k["assignments"])
    assert got.tolist() == k["expected"]
    # instances along axis 1 (kmeans.rs:428-434): a transposed view of the same data
    xt = np.asfortranarray(x)
    assert orc.update_centroids(k["centroids_shape"], xt, k["assignments"]).tolist() == k["expected"]


def test_kat_mean_squared_error(kats):
    k = kats["mean_squared_error"]
    x = np.array(k["instances"], np.float32)
    want = np.float32(k["expected_num"]) / np.float32(k["expected_den"])
    assert orc.mean_squared_error(k["centroids"], x, k["assignments"]) == want
    assert orc.mean_squared_error(k["centroids"], np.asfortranarray(x), k["assignments"]) == want


def test_update_centroids_and_loss_match_exact_model():
    K, dim, n = 5, 3, 61
    x = synth.normalish(501, (n, dim)) * np.float32(3.7)
    a = synth.codes_u8(502, (n, 1), K)[:, 0].astype(np.int64)
    a[a == 3] = 1                                    # cluster 3 stays empty -> zero centroid
    c = orc.update_centroids((K, dim), x, a)
    assert c.tobytes() == ex.update_centroids(K, x, a).tobytes()
    assert (c[3] == 0).all()
    assert orc.mean_squared_error(c, x, a) == ex.mean_squared_error(c, x, a)


def test_kmeans_iterations_compose_the_three_steps():
    """pqo_kmeans_iterations == cluster_assignments -> update_centroids -> mean_squared_error per
    subquantizer (kmeans.rs:308-327), iterated (kmeans.rs:270-279)."""
    M, K, dsub, n = 3, 8, 4, 300
    q0 = synth.normalish(511, (M, K, dsub))
    x = synth.normalish(512, (n, M * dsub))
    q, loss = q0.copy(), np.zeros(M, np.float32)
    for _ in range(3):
        for m in range(M):
            xs = x[:, m * dsub:(m + 1) * dsub]
            a = orc.cluster_assignments(q[m], np.ascontiguousarray(xs))
            q[m] = orc.update_centroids((K, dsub), xs, a)
            loss[m] = orc.mean_squared_error(q[m], xs, a)
    got_q, got_loss = orc.kmeans_iterations(q0, x, n_iterations=3)
    assert got_q.tobytes() == q.tobytes() and got_loss.tobytes() == loss.tobytes()
    got_q2, got_loss2 = orc.kmeans_iterations(q0, x, n_iterations=3, n_threads=4)
    assert got_q2.tobytes() == q.tobytes() and got_loss2.tobytes() == loss.tobytes()
    assert (got_loss[1:] != got_loss[:-1]).any()


def test_kmeans_three_spheres(kats):
    """kmeans.rs:436-480 with our own sample stream: 3 tight spheres, centroids initialised from
    instances (one per sphere is what a lucky draw gives; the reference pins its seed for that)."""
    k = kats["k_means_3"]
    centers = np.array(k["centers"], np.float32)
    pts = np.concatenate([c + np.float32(k["sigma"]) * synth.normalish(520 + i, (k["n_samples"], 2))
                          for i, c in enumerate(centers)]).astype(np.float32)
    init = pts[[3, 14, 30]][None]                    # RandomInstanceCentroids: k distinct instances
    q, loss = orc.kmeans_iterations(init, pts, n_iterations=k["iterations"])
    got = sorted(np.rint(q[0]).astype(int).tolist())
    assert got == k["expected_rounded_sorted"]
    assert loss[0] < 1e-3


# ---- OPQ training iteration without LAPACK (opq.rs:156-195) --------------------------------------
def test_at_dot_b_matches_exact_model_across_row_blocks():
    n, da, db = 530, 3, 4                       # three 256-row blocks: chain restarts + block adds
    a = synth.normalish(601, (n, da))
    b = synth.normalish(602, (n, db))
    got = orc.at_dot_b(a, b)
    want = np.array([[ex.gemm_dot(list(a[:, i]), list(b[:, j])) for j in range(db)] for i in range(da)], np.float32)
    assert got.tobytes() == want.tobytes()
    assert orc.at_dot_b(a, b, n_threads=3).tobytes() == want.tobytes()


def test_opq_train_step_composes_its_parts():
    M, K, dsub, n = 3, 8, 4, 700
    d = M * dsub
    q0 = synth.normalish(611, (M, K, dsub))
    x = synth.normalish(612, (n, d))
    P = synth.orthonormal(613, d)
    rx = orc.rotate(x, P)
    q1, _ = orc.kmeans_iterations(q0, rx, 1)
    rec = orc.reconstruct_batch(q1, orc.quantize_batch(q1, rx, dtype=np.uint64))
    cross = orc.at_dot_b(x, rec)
    got_q, got_cross = orc.opq_train_step(q0, P, x, n_threads=2)
    assert got_q.tobytes() == q1.tobytes() and got_cross.tobytes() == cross.tobytes()


def test_oracle_under_sanitizers():
    """the CPU code runs under AddressSanitizer + UBSan (GPU sanitizers are not
    available on the pool): this file's tests once more, in a child process, against the
    -fsanitize=address,undefined build of oracle/pq_oracle.c."""
    import os
    import subprocess
    import sys
    if os.environ.get("PQO_SANITIZED") == "1":
        pytest.skip("already inside the sanitized run")
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan not available")
    env = dict(os.environ, PQO_SANITIZED="1", LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_oracle.py"), "-x", "-q",
                          "-m", "not gpu", "-p", "no:cacheprovider"], env=env, cwd=root, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "passed" in out.stdout
