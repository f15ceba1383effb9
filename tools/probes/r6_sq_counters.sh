#!/bin/bash
# round 6: SQ counters (VALU issue, resident waves, waits, LDS conflicts) of the register-resident strided stage kernel against its
# LDS-resident twin (PFFT_JIT_STRIDED_HX=0): batch-interleaved N = 768 and 660 (fp32), 660 (fp64), the four-step 68640
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r6_sq; mkdir -p $out
{
echo "### register-resident (default)"
bash tools/pmc_arith.sh $out/hx bi768 768 174762 f32bi bi660 660 203360 f32bi bi660_f64 660 101680 f64bi f32_68640 68640 1920 f32
echo "### LDS-resident twin (PFFT_JIT_STRIDED_HX=0)"
PFFT_JIT_STRIDED_HX=0 bash tools/pmc_arith.sh $out/lds bi768 768 174762 f32bi bi660 660 203360 f32bi bi660_f64 660 101680 f64bi f32_68640 68640 1920 f32
} > gpurun_out/r6_sq_counters_stages.txt 2>&1
grep -v "^   GRBM\|SQ_ACTIVE" gpurun_out/r6_sq_counters_stages.txt | cut -c1-230 | head -80
