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


def _build(exe=EXE, source="facade_test.cpp"):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    src = os.path.join(ROOT, "tests", "cpp", source)
    deps = [src, os.path.join(ROOT, "include", "portfft", "portfft.hpp"), os.path.join(ROOT, "include", "portfft_amd.h"),
            os.path.join(ROOT, "portfft_amd", "libportfft_amd.so")]
    if os.path.exists(exe) and os.path.getmtime(exe) > max(os.path.getmtime(d) for d in deps):
        return
    subprocess.run([hipcc, "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), src, "-L",
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
