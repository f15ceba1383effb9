#!/bin/bash
# forced shapes of the runtime-specialised stage A of a half pair (PFFT_JIT_STRIDED_FORCE=n:fpw:lanes_per_fft:radices[:twl])
run() { echo -n "$1 | "; env PFFT_JIT_STRIDED_FORCE="$1" PFFT_JIT_VERBOSE=1 python3 tools/probes/one_desc.py ${3:-float} "$2" 10 2>&1 | grep -v amdgpu | grep "ms$\|jit\].*strided_kernel" | sed 's/.*pfa::wg_cfg/wg_cfg/' | tr '\n' ' '; echo; }
D=domain=complex,lengths=196608,batch=682
run none $D
for f in 192:16:12:16x12 192:16:16:12x16 192:16:24:8x8x3 192:16:24:8x24 192:16:8:24x8 192:16:16:16x12 192:16:12:12x16 192:16:32:6x32 192:16:24:8x6x4 192:16:12:16x12:1; do run $f $D; done
D=domain=complex,lengths=393216,batch=341
run none $D
for f in 384:16:24:16x24 384:16:16:24x16 384:16:32:12x32 384:16:48:8x8x6 384:16:32:8x8x6 384:16:24:16x6x4 384:16:48:8x6x8 384:16:64:6x8x8; do run $f $D; done
D=domain=complex,lengths=163840,batch=819
run none $D
for f in 160:16:10:16x10 160:16:16:10x16 160:16:20:8x20 160:16:8:20x8 160:16:32:5x32 160:16:20:8x5x4; do run $f $D; done
