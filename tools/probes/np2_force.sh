#!/bin/bash
# forced stage-kernel shapes of the non-power-of-two four-step plans (PFFT_JIT_STRIDED_FORCE=n:fpw:lanes_per_fft:radices[:twl])
run() { echo -n "$1 | "; env PFFT_JIT_STRIDED_FORCE="$1" python3 tools/probes/one_desc.py float "$2" 10 2>/dev/null | tail -1; }
D=domain=complex,lengths=62500,batch=2048
run none $D
for f in 250:32:25:10x5x5 250:16:25:10x5x5 250:16:50:10x5x5 250:16:25:5x5x10 250:16:50:5x10x5 250:16:25:10x25 250:16:25:25x10 250:8:25:10x5x5 250:32:50:10x5x5 250:16:10:25x10; do run $f $D; done
D=domain=complex,lengths=40000,batch=3200
run none $D
for f in 200:32:20:10x20 200:16:20:10x20 200:16:20:20x10 200:16:40:5x8x5 200:16:25:8x5x5 200:16:10:20x10; do run $f $D; done
D=domain=complex,lengths=1000000,batch=128
run none $D
for f in 1000:16:52:10x10x10 1000:16:64:10x10x10 1000:16:50:10x10x10 1000:16:40:25x40 1000:16:50:20x50 1000:16:50:10x10x10:1 1000:16:50:10x10x10:2 1000:8:100:10x10x10 1000:8:50:10x10x10 1000:16:25:40x25 1000:16:40:25x8x5 1000:16:50:20x10x5; do run $f $D; done
