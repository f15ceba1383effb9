"""The header-only C++ facade (include/portfft/portfft.hpp) compiled as user code against libportfft_amd.so.
CPU: it builds and the host-side checks (defaults, 33/17 known answer, exception types) pass.
GPU: descriptor -> commit -> compute_forward/backward -> wait against a double-precision DFT."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "build", "facade_test")


def _build():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    src = os.path.join(ROOT, "tests", "cpp", "facade_test.cpp")
    if os.path.exists(EXE) and os.path.getmtime(EXE) > max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "include", "portfft", "portfft.hpp"))):
        return
    subprocess.run([hipcc, "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), src, "-L",
                    os.path.join(ROOT, "portfft_amd"), "-lportfft_amd", "-Wl,-rpath," + os.path.join(ROOT, "portfft_amd"),
                    "-o", EXE], check=True)


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
