"""Runtime specialisation (portfft_amd/csrc/jit_planner.cpp, jit.cpp): the planner's invariants for every length up to 20000 in both
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
    assert p.stdout.count("hiprtc n=") == 24  # (round 6: + 3 lengths x 2 forms of the register-resident strided kernel, - 2 row-lanes)
    assert p.stdout.count("hiprtc nd ") == 3


def _stats(out):
    line = [ln for ln in out.splitlines() if ln.startswith("jit stats:")][-1].split()
    return int(line[3]), int(line[5])


def test_disk_cache_is_private_and_verified(tmp_path):
    """The on-disk cache of commit-time compiled kernels (jit.cpp): directory 0700, files 0600 carrying the SHA-256 of
    everything that determines the code object; a second process loads from disk; a file with a foreign digest, a
    truncated one, or one that others may write is ignored and recompiled; a group/world-writable directory is not
    used at all."""
    import stat
    _build()
    cache = tmp_path / "jit"
    env = dict(os.environ, PFFT_JIT_CACHE_DIR=str(cache))

    def run():
        p = subprocess.run([EXE, "compile"], capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0, p.stdout + p.stderr
        return _stats(p.stdout)

    compiled, from_disk = run()
    n = compiled  # one code object per compile case of jit_planner_test.cpp (packed, strided, row-lanes, hx, N-D forms)
    assert n >= 23 and from_disk == 0
    files = sorted(cache.glob("pfft_*.bin"))
    assert len(files) == n
    assert stat.S_IMODE(cache.stat().st_mode) == 0o700
    for f in files:
        assert stat.S_IMODE(f.stat().st_mode) == 0o600
        head = f.read_bytes()[:72]
        assert head[:8] == b"PFFTJIT2" and head[8:40].decode() == f.name[5:37]  # named by its own digest
    assert run() == (0, n)
    # a planted file under another kernel's name: the digest inside does not match the key -> recompiled
    files[0].write_bytes(files[1].read_bytes())
    # a truncated file
    files[2].write_bytes(files[2].read_bytes()[:100])
    # a file others may write
    files[3].chmod(0o666)
    assert run() == (3, n - 3)
    assert run() == (0, n)  # the three were rewritten (0600 again)
    # a directory others may write is not trusted at all
    cache.chmod(0o777)
    assert run() == (n, 0)
