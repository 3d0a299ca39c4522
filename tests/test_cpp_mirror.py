"""The C++ host mirror (include/reductive_amd/pq.hpp) over the C ABI: tests/cpp/test_pq_mirror.cpp
restates the reference's unit tests pq.rs:409-490.  CPU: host-side panics + single-vector path +
loud failure without a device (exit 77).  GPU: the batch path too (exit 0)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    import reductive_amd
    if not os.path.exists(reductive_amd.lib_path()):
        reductive_amd.build()
    exe = str(tmp_path / "test_pq_mirror")
    libdir = os.path.join(ROOT, "reductive_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_pq_mirror.cpp"),
                           "-L", libdir, "-lpqhip", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-o", exe])
    return exe


def test_cpp_mirror_host_checks(tmp_path):
    out = subprocess.run([_build(tmp_path)], capture_output=True, text=True)
    assert out.returncode in (0, 77), out.stdout + out.stderr


@pytest.mark.gpu
def test_cpp_mirror_on_gpu(tmp_path):
    out = subprocess.run([_build(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all checks passed (GPU)" in out.stdout
    print(out.stdout)                             # hash rate and concurrency ratio of this box (pytest -s / the log)


def _build_cache_test(tmp_path):
    import reductive_amd
    if not os.path.exists(reductive_amd.lib_path()):
        reductive_amd.build()
    from oracle import pq_oracle
    pq_oracle.build()                             # the checker of the concurrent-callers section (test infrastructure)
    exe = str(tmp_path / "test_codebook_cache")
    libdir, odir = os.path.join(ROOT, "reductive_amd"), os.path.join(ROOT, "oracle")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_codebook_cache.cpp"),
                           "-L", libdir, "-lpqhip", "-Wl,-rpath," + libdir, "-L", odir, "-lpq_oracle", "-Wl,-rpath," + odir,
                           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-o", exe])
    return exe


def test_codebook_cache_policy(tmp_path):
    """The device-codebook cache of the reference-side binding (rust/pqhip_ffi.rs mirrors
    include/reductive_amd/codebook_cache.hpp): validated hits, stale images replaced, bounded, entries pinned
    while a call runs, generation bumped by the training entry points."""
    out = subprocess.run([_build_cache_test(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "cache policy checks passed" in out.stdout


@pytest.mark.gpu
def test_codebook_cache_never_serves_a_stale_device_image(tmp_path):
    """... and 4 host threads x 2 quantizers through one cache overlap (ratio reported; guard: not fully serialised) with oracle-equal codes."""
    out = subprocess.run([_build_cache_test(tmp_path), "gpu"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all checks passed (GPU)" in out.stdout
    print(out.stdout)                             # hash rate and concurrency ratio of this box (pytest -s / the log)
