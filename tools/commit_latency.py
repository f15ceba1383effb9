"""Commit latency (descriptor.commit() wall time) per BASELINE config and per reference test size, with a cold and a
warm JIT cache: a process per measurement (the kernel tables are per process), three runs of each --
  cold   : empty PFFT_JIT_CACHE_DIR (runtime-specialised lengths compile under hiprtc; registered lengths look up)
  warm   : the same directory again (code objects read from disk)
  second : a second commit of the same descriptor inside one process (everything from the process tables)
The reference builds its kernels at commit too (SYCL specialization constants,
src/portfft/committed_descriptor_impl.hpp:448-573).  usage: commit_latency.py"""
import json, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, time, json
sys.path.insert(0, %r)
import torch, portfft_amd as pf
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
lengths = [int(v) for v in sys.argv[1].split("x")]; prec = sys.argv[2]; batch = int(sys.argv[3])
out = []
for rep in range(2):
    d = pf.descriptor(lengths, prec); d.number_of_transforms = batch
    t0 = time.perf_counter(); plan = d.commit(); torch.cuda.synchronize(); out.append((time.perf_counter() - t0) * 1e3)
print(json.dumps(out))
""" % ROOT
CASES = [("C1 fp32 N=64 b=1", "64", "f32", 1), ("C2 fp32 N=4096 b=65536", "4096", "f32", 65536),
         ("C3 fp64 N=2^20 b=128", "1048576", "f64", 128), ("C5 fp32 1024x1024 b=256", "1024x1024", "f32", 256),
         ("ref fp32 N=65536 b=2048", "65536", "f32", 2048), ("fp32 N=2^18 b=512 (XCD-local launch)", "262144", "f32", 512),
         # reference test sizes (test/unit_test/instantiate_fft_tests.hpp): registered lengths and runtime-specialised ones
         ("fp32 N=96 b=555", "96", "f32", 555), ("fp32 N=1536 b=131", "1536", "f32", 131),
         ("fp32 N=9800 b=3 (runtime-specialised)", "9800", "f32", 3), ("fp32 N=15360 b=3 (runtime-specialised)", "15360", "f32", 3),
         ("fp32 N=68640 b=3 (four-step, both stages runtime-specialised)", "68640", "f32", 3),
         ("fp32 N=3000 b=64 (runtime-specialised)", "3000", "f32", 64), ("fp64 N=5040 b=64 (runtime-specialised)", "5040", "f64", 64),
         ("fp32 104 (UNPACKED stride 3 / 4 suite length), packed here", "104", "f32", 33000),
         ("fp32 16x512 b=33", "16x512", "f32", 33), ("fp32 N=10^6 b=2 (four-step, runtime-specialised)", "1000000", "f32", 2)]
print("%-66s %10s %10s %10s   (ms: first commit of the process / second commit in the same process)" % ("", "cold", "warm", "second"))
for name, lengths, prec, batch in CASES:
    cache = tempfile.mkdtemp(prefix="pfft_cl_")
    row = []
    for mode in ("cold", "warm"):
        env = dict(os.environ, PFFT_JIT_CACHE_DIR=cache)
        p = subprocess.run([sys.executable, "-c", CHILD, lengths, prec, str(batch)], env=env, capture_output=True, text=True)
        try:
            vals = json.loads([l for l in p.stdout.splitlines() if l.startswith("[")][-1])
        except Exception:
            vals = [float("nan"), float("nan")]
        row.append(vals)
    shutil.rmtree(cache, ignore_errors=True)
    print("%-66s %10.1f %10.1f %10.2f" % (name, row[0][0], row[1][0], row[1][1]), flush=True)
