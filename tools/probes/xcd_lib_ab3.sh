# crossover batches: XCD-local plan forced (PFFT_XCD_MIN_BATCH=0) against the two-launch plan
mkdir -p gpurun_out/r4_xlib
pr() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['ms_per_step'], r['frac'], r['frac_wall'], r['kernel_ms'], r['launches_per_execute'])"; }
man() { python bench.py --manual d=cpx,n=$2,b=$3 --precision $1 --no-cpu-baseline --steps 60 2>/dev/null | pr; }
{
for spec in "float 262144 128" "float 262144 192" "float 262144 256" "float 262144 384" "float 131072 384" "float 131072 512" "float 131072 768" "float 65536 1280" "float 65536 1536" "double 65536 512" "double 65536 768" "double 131072 384" "double 262144 128" "float 1048576 128" "float 524288 384"; do
  set -- $spec
  echo -n "$1 n=$2 b=$3 xcd: "; PFFT_XCD_MIN_BATCH=0 man $1 $2 $3
  echo -n "$1 n=$2 b=$3 two: "; PFFT_NO_XCD_LOCAL=1 man $1 $2 $3
done
} 2>&1 | tee gpurun_out/r4_xlib/ab3.txt
