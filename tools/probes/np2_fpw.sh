# runtime-specialised four-step stages: group width (columns) and LDS image against work-groups per CU
pr() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(r['frac'], r['kernel'][-60:])"; }
man() { python bench.py --manual d=cpx,n=$1,b=$2 --precision float --no-cpu-baseline --steps 40 2>/dev/null | pr; }
for spec in "68640 1955" "100000 1342" "120000 1118" "250000 536" "62500 2147" "30000 4473" "40000 3355" "36000 3728" "84000 1597" "50000 2684" "200000 671"; do
  set -- $spec
  echo "n=$1"
  echo -n "  default:      "; man $1 $2
  echo -n "  fpw<=16:      "; PFFT_JIT_STRIDED_FPW=16 man $1 $2
  echo -n "  image<=80KiB: "; PFFT_JIT_STRIDED_LDS_KIB=80 man $1 $2
  echo -n "  image<=53KiB: "; PFFT_JIT_STRIDED_LDS_KIB=53 man $1 $2
done
