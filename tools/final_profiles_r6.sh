#!/bin/bash
# End-of-round artefacts (run through gpurun): bench lines of every config, rocprofv3 kernel stats and PMC traffic of the
# multi-launch / four-step configs, the survey tables, commit latency.  Copy gpurun_out/final_r6/* into profiles/ with
# cp gpurun_out/final_r6/r6_* profiles/.
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/final_r6; mkdir -p $out
ALL="bi32_2048 bi64_2048 c3 c5 ref16 ref256 ref4096 ref65536 ref9800 ref15360 ref68640 g32_14 g32_15 g32_17 g32_18 g32_19 g32_20 g32_21 g32_22 g32_24 g64_13 g64_14 g64_16 g64_17 g64_18"
PMC="c2 bi32_2048 bi64_2048 c3 c5 ref16 ref256 ref4096 ref65536 ref9800 ref15360 ref68640 g32_14 g32_15 g32_17 g32_18 g32_19 g32_20 g32_21 g32_22 g32_24 g64_13 g64_14 g64_16 g64_17 g64_18"
# ONLY_PMC="cfg ...": only the PMC passes and summaries of those configs (bench lines of them are refreshed too)
if [ -n "${ONLY_PMC:-}" ]; then PMC="$ONLY_PMC"; fi
if [ -z "${ONLY_PMC:-}" ]; then
python bench.py > $out/r6_bench_c2.json 2> $out/c2.err
for c in $ALL; do python bench.py --config $c --no-cpu-baseline > $out/r6_bench_$c.json 2> $out/$c.err; done
for c in c2 $ALL; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$c -- python3 bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline > $out/stats_$c.log 2>&1
  f=$(ls $out/stats_$c/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" $out/r6_${c}_kernel_stats.csv
  rm -rf $out/stats_$c
done
fi
for c in $PMC; do tools/run_pmc.sh $c $out/pmc_$c > $out/pmc_$c.log 2>&1; done
sum() {  # config kernels alg-bytes launches label
  case " $PMC " in *" $1 "*) ;; *) return;; esac
  local t=$out/r6_pmc_traffic_$1.json; [ $1 = c2 ] && t=$out/r6_pmc_traffic.json
  python3 tools/summarize_pmc.py $out/pmc_$1 $t --config $1 --kernels "$2" --alg-bytes $3 --launches-per-execute $4 --label "$5" > $out/pmc_$1.sum 2>&1
  python3 - $out/r6_bench_$1.json $t <<'PY'
import json, sys
b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t = json.load(open(sys.argv[2]))
t["bench_kernel_label"] = b["roofline"]["kernel"]
json.dump(t, open(sys.argv[2], "w"), indent=1)
PY
}
sum c2 stockham_wg 4294967296 1 "C2 fp32 N=4096 x 65536, one launch"
sum c3 stockham_strided 4294967296 8 "C3 fp64 N=2^20 x 128: four-step, 8 chunks of 256 MiB; FETCH_SIZE counts Infinity-Cache hits"
sum c5 stockham_rows2d,stockham_strided 4294967296 8 "C5 fp32 1024x1024 x 256: two-pass 2-D plan, 8 chunks of 256 MiB"
sum ref65536 stockham_xcd_fourstep 2147483648 1 "fp32 N=65536 x 2048: XCD-local single launch (256 x 256), slot rings of 24 transforms per XCD"
sum g32_15 stockham_wg_hx 2147483648 1 "fp32 N=32768 x 4096: register-resident work-group kernel (32.32.32 on 1024 lanes), one launch"
sum g32_14 stockham_wg_hx 2147483648 1 "fp32 N=16384 x 8192: register-resident work-group kernel, two work-groups per CU (32.32.16 on 512 lanes), one launch"
sum g64_13 stockham_wg_hx 2147483648 1 "fp64 N=8192 x 8192: register-resident work-group kernel, two work-groups per CU (16.32.16 on 256 lanes), one launch"
sum g64_14 stockham_wg_hx 2147483648 1 "fp64 N=16384 x 4096: register-resident work-group kernel (16.32.32 on 512 lanes), one launch"
sum ref16 stockham_wg 2147483648 1 "reference bench set: fp32 N=16 x 8Mi, one launch"
sum ref256 stockham_wg 2147483648 1 "reference bench set: fp32 N=256 x 512Ki, one launch"
sum ref4096 stockham_wg 2147483648 1 "reference bench set: fp32 N=4096 x 32Ki, one launch"
sum g32_17 stockham_xcd_fourstep 2147483648 1 "fp32 N=2^17 x 1024: XCD-local single launch (256 x 512), slot rings of 24 transforms per XCD"
sum g32_19 stockham_strided 2147483648 4 "fp32 N=2^19 x 256: four-step, 4 chunks of 256 MiB (the XCD-local pair of this length was de-registered in round 5)"
sum g32_18 stockham_xcd_fourstep 2147483648 1 "fp32 N=2^18 x 512: XCD-local single launch (512 x 512), slot rings of 12 transforms per XCD"
sum g32_20 stockham_strided 2147483648 4 "fp32 N=2^20 x 128: four-step, 4 chunks of 256 MiB"
sum g32_21 stockham_strided 2147483648 4 "fp32 N=2^21 x 64: four-step (1024 x 2048), 4 chunks of 256 MiB"
sum g32_22 stockham_strided 2147483648 4 "fp32 N=2^22 x 32: four-step, 4 chunks of 256 MiB"
sum g32_24 stockham_strided 2147483648 1 "fp32 N=2^24 x 8: three stages (256 x 256 x 256): 4 + 4 chunked launches of stages 1-2, one launch of stage 3"
sum ref9800 stockham_wg 2147483648 1 "fp32 N=9800 x 13312: one launch (tuned table: 7.8.7.5.5)"
sum ref15360 stockham_wg_hx 2147483648 1 "fp32 N=15360 x 8704: register-resident work-group kernel planned at commit, two work-groups per CU (32.30.16 on 512 lanes), one launch"
sum ref68640 stockham_strided 2147483648 4 "fp32 N=68640 x 1920: four-step (104 x 660, tuned table), 4 chunks of 256 MiB; stage A walks its groups XCD-contiguously, stage B is the register-resident stage kernel (660 points x 16 rows, 2 x 480 lanes per CU)"
sum g64_16 stockham_xcd_fourstep 2147483648 1 "fp64 N=65536 x 1024: XCD-local single launch (256 x 256), slot rings of 16 transforms per XCD"
sum g64_17 stockham_xcd_fourstep 2147483648 1 "fp64 N=2^17 x 512: XCD-local single launch (256 x 512), slot rings of 16 transforms per XCD"
sum g64_18 stockham_xcd_fourstep 2147483648 1 "fp64 N=2^18 x 256: XCD-local single launch (512 x 512), slot rings of 4 transforms per XCD"
sum bi32_2048 stockham_strided_hx 2147483648 1 "fp32 N=2048 x 65536 BATCH_INTERLEAVED: one pass on the one-per-CU register-resident strided kernel (16.16.8 x 16 columns on 1024 lanes, half image 128 KiB)"
sum bi64_2048 stockham_strided_hx 2147483648 1 "fp64 N=2048 x 32768 BATCH_INTERLEAVED: one pass on the one-per-CU register-resident strided kernel (16.16.8 x 8 columns on 512 lanes, half image 128 KiB)"
cp $out/r6_pmc_traffic*.json profiles/ 2>/dev/null
for c in $PMC; do
  if [ $c = c2 ]; then python bench.py > $out/r6_bench_c2.json 2> $out/c2.err; else python bench.py --config $c --no-cpu-baseline > $out/r6_bench_$c.json 2> $out/$c.err; fi
done
for c in $PMC; do f=$(ls $out/pmc_$c/stats/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && [ ! -f $out/r6_${c}_kernel_stats.csv ] && cp "$f" $out/r6_${c}_kernel_stats.csv; done
rm -rf $out/pmc_*/FETCH_SIZE $out/pmc_*/WRITE_SIZE $out/pmc_*/stats
if [ -n "${ONLY_PMC:-}" ]; then exit 0; fi
export HSA_ENABLE_IPC_MODE_LEGACY=0
PFFT_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline > $out/r6_bench_rccl_world1.json 2> $out/rccl_w1.err
PFFT_BENCH_ONE_DEVICE=1 python bench.py --gpus 8 --steps 10 --warmup 2 --no-cpu-baseline > $out/r6_bench_8rank_one_device.json 2> $out/8rank.err
python tools/commit_latency.py > $out/r6_commit_latency.txt 2>&1
tools/survey.sh $out/survey > $out/survey.log 2>&1
for f in $out/r6_bench_*.json; do python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r.get('frac_wall'), r.get('traffic'), (r.get('copy_probe') or {}).get('gbs'))" $f; done
