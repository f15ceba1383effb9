# PFFT_PLAN_MEASURE=1 on four-step lengths: the split is timed at the first commit, the record is read afterwards
export PFFT_JIT_CACHE_DIR=$(mktemp -d)
pr() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(r['frac'], r['kernel'][-62:])"; }
man() { python bench.py --manual d=cpx,n=$2,b=$3 --precision $1 --no-cpu-baseline --steps 40 2>/dev/null | pr; }
for spec in "float 68640 1955" "float 100000 1342" "float 250000 536" "float 1000000 134" "float 30000 4473" "double 68640 977" "double 100000 671" "double 30000 2236" "double 250000 268"; do
  set -- $spec
  echo "$1 n=$2"
  echo -n "  static:            "; man $1 $2 $3
  t0=$(date +%s.%N); echo -n "  measured (first):  "; PFFT_PLAN_MEASURE=1 man $1 $2 $3; t1=$(date +%s.%N)
  echo -n "  measured (record): "; PFFT_PLAN_MEASURE=1 man $1 $2 $3; t2=$(date +%s.%N)
  python -c "print('  wall: first %.1f s, with the record %.1f s' % ($t1-$t0, $t2-$t1))"
done
ls $PFFT_JIT_CACHE_DIR | grep choice | head -20; cat $PFFT_JIT_CACHE_DIR/choice_* | head -12
