# A/B of library builds / env knobs over the fp32 four-step survey rows (65536, 2^20, 10^6, 62500, 30000)
# OLD_LIB=<path to another build of libportfft_amd.so> adds that build as a third column (PORTFFT_AMD_LIBRARY)
for i in 1 2; do
for e in "A=1" "PFFT_JIT_WRITER_AUX=0x102" ${OLD_LIB:+"PORTFFT_AMD_LIBRARY=$OLD_LIB"}; do
  echo "== $e"; env $e python tools/perf_global_f32.py 2>&1 | grep TB/s | sed -n '1p;3p;5,7p' | cut -c1-75
done; done
