"""Planner experiments for the runtime-specialised kernels of the two-pass 2-D plan: one shape, many forced
configurations (PFFT_JIT_STRIDED_FORCE / PFFT_JIT_ROWS2D_FORCE, see jit.cpp), each in a child process.
usage: jit_sweep_2d.py <f32|f64> <AxB> <strided|rows2d> <force string>...      (first line: the planner's own choice)
  strided force string: n:fpw:lanes_per_fft:r0xr1x...[:twl]       rows2d: n1:rc:lanes:r0x...xr_last[:twl]"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
prec, shape, kind = sys.argv[1:4]
var = "PFFT_JIT_STRIDED_FORCE" if kind == "strided" else "PFFT_JIT_ROWS2D_FORCE"
for force in [None] + sys.argv[4:]:
    env = dict(os.environ, PFFT_JIT_VERBOSE="1")
    if force:
        env[var] = force
    p = subprocess.run([sys.executable, os.path.join(HERE, "perf_2d.py"), "child", prec, shape], env=env,
                       capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if "TB/s" in l]
    cfgs = [l.split("pfa::wg_cfg<")[1].split(">, ")[0] + ">" + l.split(">, ")[1][:14] for l in p.stderr.splitlines()
            if "[portfft_amd jit]" in l and "wg_cfg<" in l and ("stockham_" + kind) in l and "false, false" in l]
    ms = line[-1].split("ms")[0].split()[-1] if line else "FAILED " + (p.stderr or p.stdout)[-200:].replace("\n", " ")
    print("%-34s %s ms   %s" % (force or "(planner)", ms, " | ".join(cfgs[:1])), flush=True)
