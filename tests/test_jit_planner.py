"""Runtime specialisation (portfft_amd/csrc/jit.cpp): the planner's invariants for every length up to 20000 in both
precisions, and -- without a GPU -- hiprtc compilation of the embedded kernel headers for gfx950."""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "build", "jit_planner_test")


def _build():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    src = os.path.join(ROOT, "tests", "cpp", "jit_planner_test.cpp")
    lib = os.path.join(ROOT, "portfft_amd", "libportfft_amd.so")
    if os.path.exists(EXE) and os.path.getmtime(EXE) > max(os.path.getmtime(src), os.path.getmtime(lib)):
        return
    subprocess.run([hipcc, "-std=c++17", "-O1", src, "-L", os.path.join(ROOT, "portfft_amd"), "-lportfft_amd",
                    "-Wl,-rpath," + os.path.join(ROOT, "portfft_amd"), "-o", EXE], check=True)


def test_planner_invariants_and_hiprtc_compile():
    _build()
    p = subprocess.run([EXE, "compile"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "jit planner OK" in p.stdout
    assert p.stdout.count("hiprtc n=") == 5
    assert p.stdout.count("hiprtc nd ") == 3
