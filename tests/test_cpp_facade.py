"""The header-only C++ facade (include/portfft/portfft.hpp) compiled as user code against libportfft_amd.so.
CPU: it builds and the host-side checks (defaults, 33/17 known answer, exception types) pass.
GPU: descriptor -> commit -> compute_forward/backward -> wait against a double-precision DFT."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "build", "facade_test")
REF_SHAPE_EXE = os.path.join(ROOT, "build", "ref_shape_test")


MULTI_DEVICE_EXE = os.path.join(ROOT, "build", "multi_device_test")


def _build(exe=EXE, source="facade_test.cpp", extra=()):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    src = os.path.join(ROOT, "tests", "cpp", source)
    deps = [src, os.path.join(ROOT, "include", "portfft", "portfft.hpp"), os.path.join(ROOT, "include", "portfft_amd.h"),
            os.path.join(ROOT, "portfft_amd", "libportfft_amd.so")]
    if os.path.exists(exe) and os.path.getmtime(exe) > max(os.path.getmtime(d) for d in deps):
        return
    subprocess.run([hipcc, "-std=c++17", "-O1", *extra, "-I", os.path.join(ROOT, "include"), src, "-L",
                    os.path.join(ROOT, "portfft_amd"), "-lportfft_amd", "-Wl,-rpath," + os.path.join(ROOT, "portfft_amd"),
                    "-o", exe], check=True)


def test_facade_builds_and_host_checks_pass():
    _build()
    p = subprocess.run([EXE, "host"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "host checks OK" in p.stdout


@pytest.mark.gpu
def test_facade_on_gpu():
    _build()
    p = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "facade OK" in p.stdout


def test_reference_call_shapes_compile():
    """user code in the call shapes of the reference's callers (dependencies vector on all eight USM overloads, chained
    submissions, copies of a committed_descriptor) compiles against the facade"""
    _build(REF_SHAPE_EXE, "ref_shape_test.cpp")


@pytest.mark.gpu
def test_reference_call_shapes_on_gpu():
    """tests/cpp/ref_shape_test.cpp: the unit-test driver's shape (fft_test_utils.hpp:286-333), the bench loop's
    10 chained submissions (launch_bench.hpp:118-139), per-submission events, cross-stream dependencies, copies"""
    _build(REF_SHAPE_EXE, "ref_shape_test.cpp")
    p = subprocess.run([REF_SHAPE_EXE], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "ref shape OK" in p.stdout


def test_host_threads_race_through_the_host_api_and_the_runtime_compiler(tmp_path):
    """tests/cpp/multi_device_test.cpp, "host" mode (no device): four threads at once through the descriptor entry
    points, the thread-local error messages and the runtime compiler with a cold on-disk cache
    (tools/sanitize_tsan.sh runs the same under ThreadSanitizer)"""
    _build(MULTI_DEVICE_EXE, "multi_device_test.cpp", ("-pthread", "--offload-arch=gfx950"))
    env = dict(os.environ, PFFT_JIT_CACHE_DIR=str(tmp_path))
    p = subprocess.run([MULTI_DEVICE_EXE, "host"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "multi device host OK" in p.stdout


@pytest.mark.gpu
def test_one_host_thread_per_device(tmp_path):
    """per-GPU plans driven by their own host thread from ONE process -- one std::thread per visible
    device (on a 1-GPU box: 4 threads x 4 streams of device 0), concurrent commits (hiprtc + a cold disk cache + the
    XCD census), executes between barriers, sampled transforms against a host DFT, copies of every plan"""
    _build(MULTI_DEVICE_EXE, "multi_device_test.cpp", ("-pthread", "--offload-arch=gfx950"))
    env = dict(os.environ, PFFT_JIT_CACHE_DIR=str(tmp_path), PFFT_XCD_CHECK="1")
    p = subprocess.run([MULTI_DEVICE_EXE], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "multi device OK" in p.stdout
    print(p.stdout)
    # more host threads than hardware queues (streams share them): the run that caught the control block of a plan
    # being cleared by a null-stream hipMemset that had not run yet when the plan's first launch started
    p = subprocess.run([MULTI_DEVICE_EXE], capture_output=True, text=True, timeout=900, env=dict(env, MDT_THREADS="8"))
    assert p.returncode == 0, p.stdout + p.stderr
    assert "multi device OK" in p.stdout
