#!/bin/bash
# End-of-round artefacts (run through gpurun): bench lines of every config, rocprofv3 kernel stats and PMC traffic of
# the multi-launch configs, the survey tables.  Copy gpurun_out/final_r3/* into profiles/ with tools/collect_r3.sh.
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/final_r3; mkdir -p $out
python bench.py > $out/r3_bench_c2.json 2> $out/c2.err
for c in c3 c5 ref16 ref256 ref4096 ref65536 g32_15 g32_17 g32_18 g32_20 g32_21 g32_22 g32_24 g64_16; do
  python bench.py --config $c --no-cpu-baseline > $out/r3_bench_$c.json 2> $out/$c.err
done
for c in c2 c3 c5 ref65536 g32_15 g32_17 g32_18 g32_20 g32_22 g64_16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$c -- python3 bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline > $out/stats_$c.log 2>&1
  f=$(ls $out/stats_$c/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" $out/r3_${c}_kernel_stats.csv
done
for c in c2 c3 c5 ref65536; do tools/run_pmc.sh $c $out/pmc_$c > $out/pmc_$c.log 2>&1; done
python3 tools/summarize_pmc.py $out/pmc_c2 $out/r3_pmc_traffic.json --config c2 --kernels stockham_wg --alg-bytes 4294967296 --label "C2 fp32 N=4096 x 65536, one launch" > $out/pmc_c2.sum 2>&1
python3 tools/summarize_pmc.py $out/pmc_c3 $out/r3_pmc_traffic_c3.json --config c3 --kernels stockham_strided --alg-bytes 4294967296 --launches-per-execute 8 --label "C3 fp64 N=2^20 x 128: four-step, 8 chunks of 256 MiB (stage A writer policy, stage B software-pipelined tiled-input reader); FETCH_SIZE counts Infinity-Cache hits" > $out/pmc_c3.sum 2>&1
python3 tools/summarize_pmc.py $out/pmc_c5 $out/r3_pmc_traffic_c5.json --config c5 --kernels stockham_rows2d,stockham_strided --alg-bytes 4294967296 --launches-per-execute 8 --label "C5 fp32 1024x1024 x 256: two-pass 2-D plan, 8 chunks of 256 MiB" > $out/pmc_c5.sum 2>&1
python3 tools/summarize_pmc.py $out/pmc_ref65536 $out/r3_pmc_traffic_ref65536.json --config ref65536 --kernels stockham_strided --alg-bytes 2147483648 --launches-per-execute 4 --label "fp32 N=65536 x 2048: four-step, 4 chunks of 256 MiB" > $out/pmc_ref65536.sum 2>&1
# the planner's record of the launches the counters were taken on goes into the traffic files (bench.py compares it
# with its live plan: roofline.traffic_matches_plan)
for c in c2 c3 c5 ref65536; do
  t=$out/r3_pmc_traffic_$c.json; [ $c = c2 ] && t=$out/r3_pmc_traffic.json
  python3 - $out/r3_bench_$c.json $t <<'PY'
import json, sys
b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t = json.load(open(sys.argv[2]))
t["bench_kernel_label"] = b["roofline"]["kernel"]
json.dump(t, open(sys.argv[2], "w"), indent=1)
PY
done
# the bench lines again, now that the traffic files exist (their roofline.traffic then agrees with the PMC summaries)
mkdir -p profiles_tmp && cp $out/r3_pmc_traffic*.json profiles/ 2>/dev/null
for c in c2 c3 c5 ref65536; do
  if [ $c = c2 ]; then python bench.py > $out/r3_bench_c2.json 2> $out/c2.err; else python bench.py --config $c --no-cpu-baseline > $out/r3_bench_$c.json 2> $out/$c.err; fi
done
rmdir profiles_tmp 2>/dev/null
# multi-rank plumbing on one device: the RCCL branch at world size 1, and 8 ranks (gloo fall-back: RCCL refuses several
# ranks per device)
export HSA_ENABLE_IPC_MODE_LEGACY=0
PFFT_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline > $out/r3_bench_rccl_world1.json 2> $out/rccl_w1.err
PFFT_BENCH_ONE_DEVICE=1 python bench.py --gpus 8 --steps 10 --warmup 2 --no-cpu-baseline > $out/r3_bench_8rank_one_device.json 2> $out/8rank.err
tools/survey.sh $out/survey > $out/survey.log 2>&1
for f in $out/r3_bench_*.json; do python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r.get('traffic'), (r.get('copy_probe') or {}).get('gbs'))" $f; done
