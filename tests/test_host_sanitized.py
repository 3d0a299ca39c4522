"""VERDICT r1 weakness 13: the product's HOST code (strided packing, pinned staging, double-buffered
drains, row sharding over device slots, scratch leases, per-stream flags, handle lifetimes) under
AddressSanitizer + UBSan.  GPU sanitizers are not available on the pool, and none of that logic needs
a GPU: tests/mock_hip/ builds every translation unit of libpqhip with -fsanitize=address,undefined and
links it against a host-memory mock of the HIP runtime (exact-size allocations, kernel launches are
no-ops); san_driver.cpp then walks the host-resident and training entry points with the shapes that
stress the host logic.  Pass = no sanitizer report, no leak, and the promised status codes."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "tests", "mock_hip")


def test_host_logic_under_asan_ubsan_with_mock_hip():
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    build = subprocess.run(["make", "-C", MOCK, "-s", "-j8"], capture_output=True, text=True, timeout=1500)
    assert build.returncode == 0, build.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    # twice: with the library's own chunk sizes, and with 1,000-row OPQ scratch chunks so that every OPQ call of the
    # driver walks many chunks and a remainder through one lease (the chunk loop of quantize_dev_impl / reconstruct_dev_impl)
    for extra in ({}, {"PQHIP_TEST_SCRATCH_ROWS": "1000"}):
        run = subprocess.run([os.path.join(MOCK, "build", "san_driver")], capture_output=True, text=True,
                             env=dict(env, **extra), timeout=900)
        out = run.stdout + run.stderr
        assert run.returncode == 0, out[-4000:]
        assert "all checks passed" in out
        assert "AddressSanitizer" not in out and "runtime error" not in out and "LeakSanitizer" not in out, out[-4000:]


def _run_devices_mode(build_dir, env, what):
    """the multi-device driver (8 mock devices, every object tagged with its device) twice: the packing path and the
    registered zero-copy input leg"""
    for extra in ({}, {"PQHIP_HOST_ZERO_COPY": "1"}):
        run = subprocess.run([os.path.join(MOCK, build_dir, "san_driver"), "devices"], capture_output=True, text=True,
                             env=dict(env, MOCK_HIP_DEVICES="8", **extra), timeout=1500)
        out = run.stdout + run.stderr
        assert run.returncode == 0, out[-4000:]
        assert "(8 tagged devices): all checks passed" in out
        assert "MOCK-HIP DEVICE MISMATCH" not in out and what not in out and "runtime error" not in out, out[-4000:]


def test_sharder_over_eight_tagged_mock_devices_under_asan_ubsan():
    """multi-GPU readiness without a node.  The library's row sharder over EIGHT device slots of the
    mock runtime, which tags every allocation, stream and event with its device and fails the run when one is used under
    another device (its self-test proves it sees five kinds of misuse first): d = 300 and d = 768 / M = 48, ragged row
    counts, OPQ, every index width, a range error in the last shard, all slots' device entry points at once, training
    entry points on slot 7 -- under AddressSanitizer + UBSan with leak detection."""
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    build = subprocess.run(["make", "-C", MOCK, "-s", "-j8"], capture_output=True, text=True, timeout=1500)
    assert build.returncode == 0, build.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    _run_devices_mode("build", env, "AddressSanitizer")


def test_sharder_over_eight_tagged_mock_devices_under_tsan():
    """... and under ThreadSanitizer: eight shard threads, eight packing pools, the leases of eight device slots."""
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    san = "-fsanitize=thread -fno-omit-frame-pointer"
    build = subprocess.run(["make", "-C", MOCK, "-s", "-j8", "SAN=" + san, "B=build_tsan"], capture_output=True, text=True, timeout=1500)
    assert build.returncode == 0, build.stderr[-3000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1")
    _run_devices_mode("build_tsan", env, "ThreadSanitizer")


def test_host_threads_under_tsan_with_mock_hip():
    """the scratch-lease pool, the per-stream flag table and the device-slot staging shared by many host
    threads, under ThreadSanitizer (same mock HIP runtime; only the driver's multi-thread section runs)."""
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    san = "-fsanitize=thread -fno-omit-frame-pointer"
    build = subprocess.run(["make", "-C", MOCK, "-s", "-j8", "SAN=" + san, "B=build_tsan"], capture_output=True, text=True, timeout=1500)
    assert build.returncode == 0, build.stderr[-3000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1")
    for extra in ({}, {"PQHIP_TEST_SCRATCH_ROWS": "1000"}):
        run = subprocess.run([os.path.join(MOCK, "build_tsan", "san_driver"), "threads"], capture_output=True, text=True,
                             env=dict(env, **extra), timeout=900)
        out = run.stdout + run.stderr
        assert run.returncode == 0, out[-4000:]
        assert "all checks passed" in out
        assert "ThreadSanitizer" not in out, out[-4000:]
