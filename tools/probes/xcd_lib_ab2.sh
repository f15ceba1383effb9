# library A/B at several batches: the XCD-local plan against the two-launch plan of the same descriptor
mkdir -p gpurun_out/r4_xlib
pr() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['ms_per_step'], r['frac'], r['frac_wall'], r['kernel_ms'], r['launches_per_execute'], d['config']['parity_rel_l2_vs_numpy'])"; }
cfg() { python bench.py --config $1 --no-cpu-baseline --steps 100 2>/dev/null | pr; }
man() { python bench.py --manual d=cpx,n=$2,b=$3 --precision $1 --no-cpu-baseline --steps 60 2>/dev/null | pr; }
{
for c in ref65536 g32_17 g32_18 g64_16 g64_17; do
  echo -n "$c xcd: "; cfg $c
  echo -n "$c two: "; PFFT_NO_XCD_LOCAL=1 cfg $c
done
for spec in "float 524288 192" "float 524288 256" "float 524288 512" "float 1048576 192" "float 1048576 256" "double 262144 192" "double 262144 256" "double 524288 192" "double 524288 256" "double 1048576 192" "double 1048576 256" "float 65536 384" "float 65536 1024" "double 65536 256" "double 131072 256"; do
  set -- $spec
  echo -n "$1 n=$2 b=$3 xcd: "; man $1 $2 $3
  echo -n "$1 n=$2 b=$3 two: "; PFFT_NO_XCD_LOCAL=1 man $1 $2 $3
done
} 2>&1 | tee gpurun_out/r4_xlib/ab2.txt
