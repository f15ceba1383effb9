# round 5, run 7: the whole GPU suite (duration!) and SQ counters of the register-resident kernels
mkdir -p gpurun_out/r5_run7
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 2>&1 | tail -40 ) 2>&1 | tee gpurun_out/r5_run7/pytest_all.txt
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
bash tools/pmc_arith.sh gpurun_out/r5_run7/sq hx_f32_32768 32768 4096 f32 hx_f64_16384 16384 4096 f64 wg_f32_16384 16384 8192 f32 2>&1 | tee gpurun_out/r5_run7/sq.txt
